/* A stand-in for librccl.so with the nine entry points libnadm resolves (csrc/nadm_step.hip, rccl_load) -- TEST INFRASTRUCTURE for
 * tests/test_comm_first_contact.py: first contact of the step's own communicator must fail loudly, on every rank, within a
 * deadline, whatever a peer does.  Behaviour by environment:
 *   NADM_STUB_INIT_FAIL_RANK=r   ncclCommInitRank returns an error on rank r
 *   NADM_STUB_INIT_HANG_RANK=r   ncclCommInitRank never returns on rank r (what the real call does while a peer is missing)
 *   NADM_STUB_ASYNC_ERROR=1      ncclCommGetAsyncError reports a failed collective
 *   NADM_STUB_TRACE=path         append one line per entered ncclCommInitRank / ncclCommAbort ("init <rank>" / "abort")
 * Collectives succeed without moving anything.  Built by the test with gcc; never shipped, never loaded by the product. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef struct stub_comm { int rank, world; } *ncclComm_t;
typedef int ncclResult_t;          /* 0 = ncclSuccess, 2 = ncclSystemError, 7 = ncclInProgress */

static int env_rank(const char* name) { const char* v = getenv(name); return v && *v ? atoi(v) : -1; }
static void trace(const char* what, int rank) {
    const char* path = getenv("NADM_STUB_TRACE");
    if (!path || !*path) return;
    FILE* f = fopen(path, "a");
    if (!f) return;
    fprintf(f, "%s %d\n", what, rank);
    fclose(f);
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { memset(id, 0x5a, sizeof(*id)); return 0; }
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    (void)id;
    trace("init", rank);
    if (rank == env_rank("NADM_STUB_INIT_HANG_RANK")) for (;;) sleep(1000);
    if (rank == env_rank("NADM_STUB_INIT_FAIL_RANK")) return 2;
    *comm = (ncclComm_t)malloc(sizeof(**comm));
    (*comm)->rank = rank; (*comm)->world = nranks;
    return 0;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { free(comm); return 0; }
ncclResult_t ncclCommAbort(ncclComm_t comm) { trace("abort", comm ? comm->rank : -1); free(comm); return 0; }
ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t* state) {
    (void)comm;
    const char* v = getenv("NADM_STUB_ASYNC_ERROR");
    *state = (v && *v == '1') ? 2 : 0;
    return 0;
}
ncclResult_t ncclReduceScatter(const void* s, void* r, size_t n, int dt, int op, ncclComm_t c, void* st) { (void)s; (void)r; (void)n; (void)dt; (void)op; (void)c; (void)st; return 0; }
ncclResult_t ncclAllGather(const void* s, void* r, size_t n, int dt, ncclComm_t c, void* st) { (void)s; (void)r; (void)n; (void)dt; (void)c; (void)st; return 0; }
ncclResult_t ncclAllReduce(const void* s, void* r, size_t n, int dt, int op, ncclComm_t c, void* st) { (void)s; (void)r; (void)n; (void)dt; (void)op; (void)c; (void)st; return 0; }
const char* ncclGetErrorString(ncclResult_t r) { return r == 0 ? "no error" : "stub: unhandled system error"; }
