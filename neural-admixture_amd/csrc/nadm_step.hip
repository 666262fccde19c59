// The training step as ONE host call (include/nadm.h, "The training step as ONE call"): nadm_plan_* / nadm_step / nadm_comm_*.
//
// Host code only -- the launches are the entry points of nadm_genotype_passes.hip / nadm_small_kernels.hip in the order
// NeuralAdmixture._run_step runs its pieces (neural_admixture.py:403-414): gather + decode + X.V, RMSNorm/MLP/softmax, decoder +
// BCE forward/backward per head, MLP backward, dV, then (one GPU) Adam + restrict_P in the epilogues of the passes that complete
// a gradient, or (several GPUs) the gradient exchange the reference leaves to DistributedDataParallel (:315-319) followed by the
// optimizer on this rank's share.  The host queues a step in one call instead of seven ctypes launches + Python collectives.
//
// WHO OWES WHICH UPDATE (the only state a step leaves behind; nadm_plan_flush settles both):
//   small_pending  single / SNP mode, C <= 8: sum of the MLP weight-gradient partials + Adam on the small parameters of step t ride
//                  as side blocks of step t+1's pass 1 (pass 1 reads no small parameter; the MLP forward behind it reads the new ones)
//   a_pending      DP mode: message A of step t (reduce-scatter of dP -> Adam + restrict_P on this rank's slice -> all-gather of P)
//                  runs on the side stream; step t+1's pass 2 -- the first reader of P -- waits for its event
//   b_pending      DP mode with n_buckets > 1: message B of step t (small | V) runs bucket by bucket on the second side stream; part j of
//                  step t+1's pass 1 -- the first reader of V's range j -- waits for bucket j's event (part 0 also brings the small
//                  parameters).  One bucket (the default): message B is on the compute stream, nothing is pending
// Everything else (P / V in single and SNP mode) is final when the step's last launch on `stream` is.
// A step that fails part-way POISONS the plan (include/nadm.h): nothing below tries to unwind a half-queued step.
#include "../../include/nadm.h"
#include "nadm_host.h"
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and prototypes only: the functions are resolved with dlsym (no link-time dependency)
#include <stdlib.h>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

using namespace nadm;

#define HIP_OK(call, what)                                                                       \
    do {                                                                                         \
        const hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                                  \
            snprintf(err_buf(), 512, "%s: %s", what, hipGetErrorString(e_));                     \
            return 3;                                                                            \
        }                                                                                        \
    } while (0)

// ------------------------------------------------------------------------------------------------- flat layout
static int64_t round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
static int64_t gcd64(int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; }
static int64_t lcm64(int64_t a, int64_t b) { return a / gcd64(a, b) * b; }

extern "C" int nadm_flat_layout(const nadm_heads_t* hd, int64_t M, int32_t world, int32_t n_buckets, nadm_flat_layout_t* out) {
    if (!hd || !out) return fail("nadm_flat_layout: null pointer");
    if (M <= 0 || world < 1 || hd->n_heads < 1 || hd->n_heads > NADM_MAX_HEADS) return fail("nadm_flat_layout: M, world >= 1 and 1..32 heads");
    if (n_buckets < 0 || n_buckets > NADM_MAX_BUCKETS) return fail("nadm_flat_layout: at most 8 buckets");
    memset(out, 0, sizeof(*out));
    const int64_t q = 4 * (int64_t)world;                               // a rank's slice is a multiple of 4 floats (16 bytes: the Adam kernel's vector accesses)
    out->off_v = round_up(hd->n_small, lcm64(64, q));
    // range boundaries of message B's buckets: multiples of U SNPs = a multiple of every pass's chunk (pass 1: 2048, pass 3: 512) whose
    // V rows are a multiple of q floats, so that every bucket but the last is `world` slices without a gap
    const int64_t U = lcm64(2048, q / gcd64(q, hd->CP));
    const int64_t units = (M + U - 1) / U;
    int64_t nb = n_buckets < 1 ? 1 : n_buckets;
    if (nb > units) nb = units;
    out->n_buckets = (int32_t)nb;
    const int64_t b_end = round_up(out->off_v + M * hd->CP, q);
    for (int64_t j = 0; j <= nb; ++j) {
        const int64_t m = j == nb ? M : U * (j * units / nb);
        out->bkt_m0[j] = m;
        out->bkt_off[j] = j == 0 ? 0 : (j == nb ? b_end : out->off_v + m * hd->CP);
    }
    for (int64_t j = 0; j < nb; ++j) {
        out->bkt_slice[j] = (out->bkt_off[j + 1] - out->bkt_off[j]) / world;
        out->bkt_mom[j] = out->slice_b;
        out->slice_b += out->bkt_slice[j];
    }
    out->msg_a_off = b_end;
    int64_t off = out->msg_a_off;
    for (int h = 0; h < hd->n_heads; ++h) {
        out->off_p[h] = off;
        off += M * hd->kp[h];
    }
    out->slice_a = round_up(off - out->msg_a_off, q) / world;
    out->n_flat = out->msg_a_off + out->slice_a * world;
    return 0;
}

// ------------------------------------------------------------------------------------------------- RCCL transport
namespace {

struct RcclFns {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;       // optional: without it a comm reports no asynchronous errors
    decltype(&ncclCommAbort) CommAbort = nullptr;                       // optional: nadm_comm_abort then falls back to leaking the communicator
};

int rccl_load(const char* path, RcclFns* f) {
    const char* name = (path && *path) ? path : "librccl.so";
    f->lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!f->lib) {
        snprintf(err_buf(), 512, "nadm_comm_rccl: cannot load %s: %s", name, dlerror());
        return 1;
    }
#define NADM_SYM(field, sym)                                                              \
    f->field = reinterpret_cast<decltype(f->field)>(dlsym(f->lib, sym));                  \
    if (!f->field) { snprintf(err_buf(), 512, "nadm_comm_rccl: %s lacks %s", name, sym); return 1; }
    NADM_SYM(GetUniqueId, "ncclGetUniqueId")
    NADM_SYM(CommInitRank, "ncclCommInitRank")
    NADM_SYM(CommDestroy, "ncclCommDestroy")
    NADM_SYM(ReduceScatter, "ncclReduceScatter")
    NADM_SYM(AllGather, "ncclAllGather")
    NADM_SYM(AllReduce, "ncclAllReduce")
    NADM_SYM(GetErrorString, "ncclGetErrorString")
#undef NADM_SYM
    f->CommGetAsyncError = reinterpret_cast<decltype(f->CommGetAsyncError)>(dlsym(f->lib, "ncclCommGetAsyncError"));
    f->CommAbort = reinterpret_cast<decltype(f->CommAbort)>(dlsym(f->lib, "ncclCommAbort"));
    return 0;
}

struct RcclCtx {
    RcclFns f;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

int rccl_fail(RcclCtx* c, const char* what, ncclResult_t r) {
    snprintf(err_buf(), 512, "%s: %s", what, c->f.GetErrorString(r));
    return 4;
}

int rccl_reduce_scatter(void* ctx, float* buf, int64_t slice, void* stream) {
    RcclCtx* c = static_cast<RcclCtx*>(ctx);
    const ncclResult_t r = c->f.ReduceScatter(buf, buf + (int64_t)c->rank * slice, (size_t)slice, ncclFloat, ncclSum, c->comm, (hipStream_t)stream);
    return r == ncclSuccess ? 0 : rccl_fail(c, "ncclReduceScatter", r);
}
int rccl_all_gather(void* ctx, float* buf, int64_t slice, void* stream) {
    RcclCtx* c = static_cast<RcclCtx*>(ctx);
    const ncclResult_t r = c->f.AllGather(buf + (int64_t)c->rank * slice, buf, (size_t)slice, ncclFloat, c->comm, (hipStream_t)stream);
    return r == ncclSuccess ? 0 : rccl_fail(c, "ncclAllGather", r);
}
int rccl_all_reduce(void* ctx, float* buf, int64_t n, void* stream) {
    RcclCtx* c = static_cast<RcclCtx*>(ctx);
    const ncclResult_t r = c->f.AllReduce(buf, buf, (size_t)n, ncclFloat, ncclSum, c->comm, (hipStream_t)stream);
    return r == ncclSuccess ? 0 : rccl_fail(c, "ncclAllReduce", r);
}
int rccl_async_error(void* ctx) {
    RcclCtx* c = static_cast<RcclCtx*>(ctx);
    if (!c->f.CommGetAsyncError || !c->comm) return 0;
    ncclResult_t state = ncclSuccess;
    const ncclResult_t r = c->f.CommGetAsyncError(c->comm, &state);
    if (r != ncclSuccess) return rccl_fail(c, "ncclCommGetAsyncError", r);
    if (state != ncclSuccess && state != ncclInProgress) {
        snprintf(err_buf(), 512, "a collective of rank %d / %d failed asynchronously: %s", c->rank, c->world, c->f.GetErrorString(state));
        return 4;
    }
    return 0;
}
void rccl_destroy(void* ctx) {
    RcclCtx* c = static_cast<RcclCtx*>(ctx);
    if (c->comm) c->f.CommDestroy(c->comm);
    delete c;                                                   // (the library stays loaded: other communicators may use it)
}

int noop_slices(void*, float*, int64_t, void*) { return 0; }

// ncclCommInitRank returns when every rank has entered it; a peer that died before doing so would keep this rank inside it
// forever.  The call runs on a helper thread (bound to the caller's HIP device) and the caller waits for it with a deadline.
struct InitJob {
    RcclFns f;
    ncclUniqueId id;
    int rank = 0, world = 1, device = -1;
    ncclComm_t comm = nullptr;
    ncclResult_t res = ncclSuccess;
    bool done = false;
    std::mutex mu;
    std::condition_variable cv;
};

}  // namespace

extern "C" int nadm_comm_rccl_probe(const char* librccl_path) {
    RcclFns f;
    return rccl_load(librccl_path, &f);
}

extern "C" int nadm_comm_rccl_unique_id(const char* librccl_path, void* id128) {
    if (!id128) return fail("nadm_comm_rccl_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) == 128, "nadm.h promises 128 bytes");
    RcclFns f;
    if (rccl_load(librccl_path, &f)) return 1;
    ncclUniqueId id;
    const ncclResult_t r = f.GetUniqueId(&id);
    if (r != ncclSuccess) {
        snprintf(err_buf(), 512, "ncclGetUniqueId: %s", f.GetErrorString(r));
        return 4;
    }
    memcpy(id128, &id, sizeof(id));
    return 0;
}

extern "C" int nadm_comm_rccl(const char* librccl_path, const void* id128, int32_t rank, int32_t world, int32_t timeout_ms, nadm_comm_t** out) {
    if (!id128 || !out) return fail("nadm_comm_rccl: null pointer");
    if (world < 1 || rank < 0 || rank >= world) return fail("nadm_comm_rccl: need 0 <= rank < world");
    auto job = std::make_shared<InitJob>();
    if (rccl_load(librccl_path, &job->f)) return 1;
    job->rank = rank; job->world = world;
    memcpy(&job->id, id128, sizeof(job->id));
    if (hipGetDevice(&job->device) != hipSuccess) job->device = -1;          // (no GPU: a test double of the library)
    std::thread([job] {
        if (job->device >= 0) (void)hipSetDevice(job->device);              // the communicator binds to the device that is current HERE
        ncclComm_t comm = nullptr;
        const ncclResult_t r = job->f.CommInitRank(&comm, job->world, job->id, job->rank);     // blocks until every rank has called it
        std::lock_guard<std::mutex> g(job->mu);
        job->comm = comm; job->res = r; job->done = true;
        job->cv.notify_all();
    }).detach();
    {
        std::unique_lock<std::mutex> lk(job->mu);
        if (timeout_ms > 0) {
            if (!job->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return job->done; })) {
                snprintf(err_buf(), 512,
                         "ncclCommInitRank: rank %d of %d gave up after %d ms -- a peer never entered the call (it failed earlier or is gone); the "
                         "helper thread stays inside the library, tear the process down",
                         rank, world, timeout_ms);
                return 5;
            }
        } else {
            job->cv.wait(lk, [&] { return job->done; });
        }
    }
    RcclCtx* c = new RcclCtx;
    c->f = job->f; c->rank = rank; c->world = world; c->comm = job->comm;
    if (job->res != ncclSuccess) {
        const int rc = rccl_fail(c, "ncclCommInitRank", job->res);
        delete c;
        return rc;
    }
    nadm_comm_t* k = new nadm_comm_t;
    k->rank = rank; k->world = world; k->ctx = c;
    k->reduce_scatter = rccl_reduce_scatter; k->all_gather = rccl_all_gather; k->all_reduce = rccl_all_reduce; k->destroy = rccl_destroy;
    k->async_error = rccl_async_error;
    *out = k;
    return 0;
}

extern "C" int nadm_comm_emulated(int32_t world, nadm_comm_t** out) {
    if (!out || world < 1) return fail("nadm_comm_emulated: world >= 1");
    nadm_comm_t* k = new nadm_comm_t;
    k->rank = 0; k->world = world; k->ctx = nullptr;
    k->reduce_scatter = noop_slices; k->all_gather = noop_slices; k->all_reduce = noop_slices; k->destroy = nullptr; k->async_error = nullptr;
    *out = k;
    return 0;
}

extern "C" void nadm_comm_free(nadm_comm_t* comm) {
    if (!comm) return;
    if (comm->destroy) comm->destroy(comm->ctx);
    delete comm;
}

extern "C" void nadm_comm_abort(nadm_comm_t* comm) {
    if (!comm) return;
    if (comm->destroy == rccl_destroy) {
        RcclCtx* c = static_cast<RcclCtx*>(comm->ctx);
        if (c->comm && c->f.CommAbort) (void)c->f.CommAbort(c->comm);     // (without ncclCommAbort the communicator is leaked: a destroy could hang)
        delete c;
    } else if (comm->destroy) {
        comm->destroy(comm->ctx);
    }
    delete comm;
}

// ------------------------------------------------------------------------------------------------- the plan
struct nadm_plan {
    nadm_plan_desc_t d;
    nadm_flat_layout_t lay;
    int rank = 0, world = 1;
    int step_count = 0;
    bool p_unit = true;
    bool poisoned = false;                                       // a step failed part-way: see include/nadm.h
    // what the last step left to the next one (see the head of this file)
    bool small_pending = false;
    int pend_splits = 0, pend_step = 0;
    float pend_lr = 0.f, pend_scale = 1.f;
    bool a_pending = false, b_pending = false;
    int last_b = 0;                                              // batch size of the previous step (dZ image hygiene, see nadm_step)
    // DP: message A on `side`, message B bucket by bucket on `side_b`; the parts 1.. of a pass 1 launched in parts on `fan`
    int nb = 1;                                                  // buckets of message B
    const nadm_comm_t* comm_a = nullptr;                         // the communicator message A travels on (d.comm unless the caller gave it its own)
    hipStream_t side = nullptr, side_b = nullptr, fan[NADM_MAX_BUCKETS] = {nullptr};
    hipEvent_t ev_p2 = nullptr, ev_a = nullptr, ev_fork1 = nullptr;
    hipEvent_t ev_p1[NADM_MAX_BUCKETS] = {nullptr};              // part j of pass 1 done (fan -> compute stream)
    hipEvent_t ev_b[NADM_MAX_BUCKETS] = {nullptr};               // range j of pass 3 done: bucket j's gradients are complete
    hipEvent_t ev_g[NADM_MAX_BUCKETS] = {nullptr};               // bucket j all-gathered: small (j = 0) and V's range j are final
    // multi-head models: the heads' pass-2 launches are independent (own P rows, slab, loss slots; they share X and Q) and each ends
    // in a partly filled round of blocks: two launches in flight fill each other's tails (r02: 3.12 -> 2.9 ms at nine heads)
    int fan_heads = 1;
    hipStream_t head_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    const int32_t* labels = nullptr;
    int n_classes = 0;
    float sup_weight = 0.f;
    int64_t enc_chunks = 0, dec_chunks[NADM_MAX_HEADS] = {0}, loss_off[NADM_MAX_HEADS] = {0}, slab_off[NADM_MAX_HEADS] = {0}, n_loss = 0;
    int32_t p3_slices_cap = 1;                      // sample slices p3_slab was sized for
    int32_t slices_cap[NADM_MAX_HEADS] = {0};       // sample slices the head's region of p2_slab was sized for (nadm_decode_slices_max at creation)
    uint32_t tmask = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> trec[NADM_T_COUNT], trec_bkt[NADM_MAX_BUCKETS];
    std::vector<hipEvent_t> pool;
};

namespace {

struct Timed {                                                   // RAII-free: begin() / end() around a group of launches on one stream
    nadm_plan* p;
    int slot;
    hipStream_t st;
    std::vector<std::pair<hipEvent_t, hipEvent_t>>* into = nullptr;   // default: the slot's own record list
    hipEvent_t a = nullptr, b = nullptr;
    int begin() {
        if (!(p->tmask & (1u << slot))) return 0;
        for (hipEvent_t* e : {&a, &b}) {
            if (!p->pool.empty()) { *e = p->pool.back(); p->pool.pop_back(); }
            else HIP_OK(hipEventCreate(e), "hipEventCreate");
        }
        HIP_OK(hipEventRecord(a, st), "hipEventRecord");
        return 0;
    }
    int end(hipStream_t on = nullptr) {                          // (a span may end on another stream than it began on)
        if (!a) return 0;
        HIP_OK(hipEventRecord(b, on ? on : st), "hipEventRecord");
        (into ? *into : p->trec[slot]).emplace_back(a, b);
        return 0;
    }
};

float* P_of(nadm_plan* p, float* flat, int h) { return flat + p->lay.off_p[h]; }

nadm_adam_t adam_at(nadm_plan* p, int64_t off, float lr, int step, float scale) {
    return nadm_adam_t{p->d.m + off, p->d.v + off, lr, step, scale, 0};
}

int flush_small(nadm_plan* p, void* stream) {
    if (!p->small_pending) return 0;
    const nadm_adam_t sa = adam_at(p, 0, p->pend_lr, p->pend_step, p->pend_scale);
    if (nadm_small_grads(p->d.small_part, p->pend_splits, p->d.heads.n_small, p->d.grads, p->d.params, &sa, stream)) return 1;
    p->small_pending = false;                                    // only now: a refused launch must not lose the update
    return 0;
}

// DP mode: pass 1 in the parts message B's buckets cut V into.  Part j reads V's range j only, so it waits for bucket j of the
// previous step only (part 0, on the compute stream, also makes the small parameters final for the MLP forward behind it); the
// parts 1.. run on streams of their own beside it and join the compute stream.  One bucket: one launch, one wait.
int encode_fwd_parts(nadm_plan* p, const int32_t* idx, int b, hipStream_t st) {
    const nadm_plan_desc_t& d = p->d;
    const nadm_heads_t& hd = d.heads;
    const float* V = d.params + p->lay.off_v;
    if (p->nb > 1) HIP_OK(hipEventRecord(p->ev_fork1, st), "hipEventRecord");     // the parts may not overtake what the compute stream still holds (zpart's readers, idx)
    for (int j = 0; j < p->nb; ++j) {
        hipStream_t fs = j == 0 ? st : p->fan[j];
        if (j > 0) HIP_OK(hipStreamWaitEvent(fs, p->ev_fork1, 0), "hipStreamWaitEvent");
        if (p->b_pending) HIP_OK(hipStreamWaitEvent(fs, p->ev_g[j], 0), "hipStreamWaitEvent");
        const int64_t m0 = p->lay.bkt_m0[j], m1 = p->lay.bkt_m0[j + 1];
        if (p->nb == 1) {
            if (nadm_encode_fwd(d.xp, d.ld, idx, b, d.M, V, hd.CP, d.zpart, fs)) return 1;
        } else if (nadm_encode_fwd_part(d.xp + m0 / 4, d.ld, idx, b, m1 - m0, V + m0 * hd.CP, hd.CP, d.zpart + nadm_encode_chunks(m0) * (int64_t)b * hd.CP,
                                        p->enc_chunks, fs)) {
            return 1;
        }
        if (j > 0) HIP_OK(hipEventRecord(p->ev_p1[j], fs), "hipEventRecord");
    }
    for (int j = 1; j < p->nb; ++j) HIP_OK(hipStreamWaitEvent(st, p->ev_p1[j], 0), "hipStreamWaitEvent");
    p->b_pending = false;                                        // the compute stream has (transitively) waited for every bucket
    return 0;
}

// pass 1 (+ the small update the previous step left to it) and the MLP forward; SNP mode: the partial Z summed over ranks in between
int forward(nadm_plan* p, const int32_t* idx, int b, void* stream) {
    const nadm_plan_desc_t& d = p->d;
    const nadm_heads_t& hd = d.heads;
    float* V = d.params + p->lay.off_v;
    Timed t1{p, NADM_T_ENCODE_FWD, (hipStream_t)stream};
    if (t1.begin()) return 1;
    if (d.mode == NADM_MODE_DP) {
        if (encode_fwd_parts(p, idx, b, (hipStream_t)stream)) return 1;
    } else if (p->small_pending && hd.CP <= 8) {
        const nadm_adam_t sa = adam_at(p, 0, p->pend_lr, p->pend_step, p->pend_scale);
        if (nadm_encode_fwd_small(d.xp, d.ld, idx, b, d.M, V, hd.CP, d.zpart, d.small_part, p->pend_splits, hd.n_small, d.grads, d.params, &sa, stream))
            return 1;
        p->small_pending = false;
    } else {
        if (flush_small(p, stream)) return 1;
        if (nadm_encode_fwd(d.xp, d.ld, idx, b, d.M, V, hd.CP, d.zpart, stream)) return 1;
    }
    if (t1.end()) return 1;
    const float* zsrc = d.zpart;
    int64_t zrows = p->enc_chunks;
    if (d.mode == NADM_MODE_SNP) {
        if (nadm_sum_rows(d.zpart, p->enc_chunks, (int64_t)b * hd.CP, d.zsum, stream)) return 1;
        if (d.comm && d.comm->all_reduce(d.comm->ctx, d.zsum, (int64_t)b * hd.CP, stream)) return 1;
        zsrc = d.zsum; zrows = 1;
    }
    Timed t2{p, NADM_T_MLP_FWD, (hipStream_t)stream};
    if (t2.begin()) return 1;
    const int rc = d.qimg ? nadm_mlp_fwd_images(&hd, d.params, zsrc, zrows, b, d.Z, d.rinv, d.Zn, d.H, d.Q, d.qimg, d.qimg_head_bytes, stream)
                          : nadm_mlp_fwd(&hd, d.params, zsrc, zrows, b, d.Z, d.rinv, d.Zn, d.H, d.Q, stream);
    if (rc) return 1;
    return t2.end();
}

// pass 2 for every head; `adam` != nullptr: Adam + restrict_P in the epilogue (lr, step, scale), else the gradient is written to grads
int decode_heads(nadm_plan* p, const int32_t* idx, int b, int with_loss, const float* lr_scale /* [lr, scale] or nullptr */, void* stream) {
    const nadm_plan_desc_t& d = p->d;
    const nadm_heads_t& hd = d.heads;
    const int flags = with_loss ? (p->p_unit ? 1 : 3) : 0;
    hipStream_t main = (hipStream_t)stream;
    const int fan = hd.n_heads > 1 ? p->fan_heads : 1;
    if (fan > 1) {
        HIP_OK(hipEventRecord(p->ev_fork, main), "hipEventRecord");
        HIP_OK(hipStreamWaitEvent(p->head_stream, p->ev_fork, 0), "hipStreamWaitEvent");
    }
    int64_t dq_off = 0;
    for (int h = 0; h < hd.n_heads; ++h) {
        const int kp = hd.kp[h];
        void* st = (fan > 1 && (h & 1)) ? (void*)p->head_stream : stream;
        float* Ph = P_of(p, d.params, h);
        float* dPh = P_of(p, d.grads, h);
        float* slab = d.dqpart + dq_off;
        float* lossp = d.losspart + p->loss_off[h];
        const float* Qh = d.Q + hd.qoff[h];
        uint8_t* xg = (h == 0 && hd.CP <= 8) ? d.xg : nullptr;          // head 0's launch leaves the batch copy for pass 3
        nadm_adam_t ad;
        const nadm_adam_t* adp = nullptr;
        if (lr_scale) { ad = adam_at(p, p->lay.off_p[h], lr_scale[0], p->step_count, lr_scale[1]); adp = &ad; }
        int rc;
        const void* qi = (d.qimg && kp <= 16) ? (const char*)d.qimg + (int64_t)h * d.qimg_head_bytes : nullptr;
        const int slices = d.p2_slab ? nadm_decode_slices(b, d.M, kp) : 1;     // sample slices where the SNP chunks alone leave CUs idle
        if (slices > 1 && slices > p->slices_cap[h])                           // (the rule is a function of (b, M, kp); only the test build can move it)
            return fail("nadm_step: pass 2 would be cut into more sample slices than the plan's slab was sized for at creation");
        if (slices > 1)
            rc = nadm_decode_bce_sliced(d.xp, d.ld, idx, b, d.M, Ph, kp, Qh, hd.SP, dPh, slab, lossp, flags, xg, adp, qi, slices,
                                        d.p2_slab + p->slab_off[h], d.p2_cnt + p->loss_off[h], st);
        else if (qi)
            rc = nadm_decode_bce_images(d.xp, d.ld, idx, b, d.M, Ph, kp, Qh, hd.SP, dPh, slab, lossp, flags, xg, adp, qi, st);
        else
            rc = nadm_decode_bce_step(d.xp, d.ld, idx, b, d.M, Ph, kp, Qh, hd.SP, dPh, slab, lossp, flags, xg, adp, st);
        if (rc) return 1;
        dq_off += p->dec_chunks[h] * (int64_t)b * kp;
    }
    if (fan > 1) {
        HIP_OK(hipEventRecord(p->ev_join, p->head_stream), "hipEventRecord");
        HIP_OK(hipStreamWaitEvent(main, p->ev_join, 0), "hipStreamWaitEvent");
    }
    return 0;
}

// one message (or bucket) of the sample-sharded step = reduce-scatter of the gradients -> Adam (+ restrict_P) on this rank's slice ->
// all-gather of the updated parameters, as three calls: message A issues them at different points of the step (nadm_step).  msg_off /
// slice in floats of the flat buffers; mom_off = where the slice's moments start in d.m / d.v
int msg_reduce(nadm_plan* p, const nadm_comm_t* c, int64_t msg_off, int64_t slice, void* stream) {
    return (c && c->reduce_scatter(c->ctx, p->d.grads + msg_off, slice, stream)) ? 1 : 0;
}
int msg_update(nadm_plan* p, int64_t msg_off, int64_t slice, int64_t mom_off, bool clamp, float lr, void* stream) {
    const nadm_plan_desc_t& d = p->d;
    const int64_t lo = msg_off + (int64_t)p->rank * slice;
    return nadm_adam(d.params + lo, d.grads + lo, d.m + mom_off, d.v + mom_off, slice, clamp ? 0 : slice, lr, p->step_count, 1.0f / (float)p->world, stream);
}
int msg_gather(nadm_plan* p, const nadm_comm_t* c, int64_t msg_off, int64_t slice, void* stream) {
    return (c && c->all_gather(c->ctx, p->d.params + msg_off, slice, stream)) ? 1 : 0;
}

int comm_health(const nadm_plan* p) {
    for (const nadm_comm_t* c : {p->d.comm, p->comm_a != p->d.comm ? p->comm_a : nullptr})
        if (c && c->async_error && c->async_error(c->ctx)) return 1;
    return 0;
}

int poisoned(const nadm_plan* p, const char* who) {
    snprintf(err_buf(), 512, "%s: an earlier nadm_step on this plan failed part-way; its state is undefined -- destroy the plan and build a new one", who);
    return 6;
}

}  // namespace

extern "C" int nadm_plan_create(const nadm_plan_desc_t* desc, nadm_plan_t** out) {
    if (!desc || !out) return fail("nadm_plan_create: null pointer");
    const nadm_plan_desc_t& d = *desc;
    if (d.mode != NADM_MODE_SINGLE && d.mode != NADM_MODE_DP && d.mode != NADM_MODE_SNP) return fail("nadm_plan_create: unknown mode");
    if (d.M <= 0 || d.bmax <= 0 || d.ld % 16 != 0 || d.ld * 4 < d.M) return fail("nadm_plan_create: M, bmax > 0; ld a multiple of 16 and >= ceil(M/4)");
    if (d.ld >> 32) return fail("nadm_plan_create: rows of 4 GiB and more (ld >= 2^32) are not supported");      // the genotype passes form row addresses from 32-bit factors
    const nadm_heads_t& hd = d.heads;
    if (hd.n_heads < 1 || hd.n_heads > NADM_MAX_HEADS) return fail("nadm_plan_create: head table not initialised (nadm_heads_init)");
    if (!d.params || !d.grads || !d.m || !d.v || !d.zpart || !d.Z || !d.rinv || !d.Zn || !d.H || !d.Q || !d.dL || !d.dHpre || !d.dgp || !d.dZ ||
        !d.dqpart || !d.losspart || !d.small_part || !d.loss_acc)
        return fail("nadm_plan_create: null pointer in the descriptor");
    if (hd.CP <= 8 && (!d.dzimg || !d.dzcnt || !d.xg)) return fail("nadm_plan_create: C <= 8 needs dzimg, dzcnt and xg");
    if (d.mode == NADM_MODE_SNP && (!d.zsum || !d.dqsum)) return fail("nadm_plan_create: SNP mode needs zsum and dqsum");
    if (d.mode == NADM_MODE_SINGLE && d.comm && d.comm->world != 1) return fail("nadm_plan_create: single mode with a communicator of several ranks");
    for (const nadm_comm_t* c : {d.comm, d.comm_a})
        if (c && (!c->reduce_scatter || !c->all_gather || !c->all_reduce || c->world < 1 || c->rank < 0 || c->rank >= c->world))
            return fail("nadm_plan_create: incomplete communicator");
    if (d.comm_a && (d.mode != NADM_MODE_DP || !d.comm || d.comm_a->world != d.comm->world || d.comm_a->rank != d.comm->rank))
        return fail("nadm_plan_create: comm_a is the second communicator of the sample-sharded mode: same rank and world as comm");
    if (d.n_buckets < 0 || d.n_buckets > NADM_MAX_BUCKETS) return fail("nadm_plan_create: at most 8 buckets");
    if (d.reserved != 0) return fail("nadm_plan_create: nadm_plan_desc_t.reserved must be 0");
    if ((d.p2_slab == nullptr) != (d.p2_cnt == nullptr) || ((uintptr_t)d.p2_slab & 15))
        return fail("nadm_plan_create: p2_slab (16-byte aligned) and p2_cnt come together (both NULL: pass 2 is never sliced)");
    if ((d.p3_slab == nullptr) != (d.p3_cnt == nullptr) || ((uintptr_t)d.p3_slab & 15))
        return fail("nadm_plan_create: p3_slab (16-byte aligned) and p3_cnt come together (both NULL: pass 3 is never sliced)");
    nadm_plan* p = new nadm_plan;
    p->d = d;
    p->world = d.comm ? d.comm->world : 1;
    p->rank = d.comm ? d.comm->rank : 0;
    p->comm_a = d.comm_a ? d.comm_a : d.comm;
    const bool dp = d.mode == NADM_MODE_DP;
    if (nadm_flat_layout(&hd, d.M, dp ? p->world : 1, dp ? d.n_buckets : 1, &p->lay)) { delete p; return 1; }
    p->nb = p->lay.n_buckets;
    p->enc_chunks = nadm_encode_chunks(d.M);
    int64_t slab_floats = 0;
    for (int h = 0; h < hd.n_heads; ++h) {
        p->dec_chunks[h] = nadm_decode_chunks(d.M, hd.kp[h]);
        p->loss_off[h] = p->n_loss;                              // (= the head's offset into the slice counters as well)
        p->n_loss += p->dec_chunks[h];
        p->slab_off[h] = slab_floats;
        p->slices_cap[h] = nadm_decode_slices_max(d.bmax, d.M, hd.kp[h]);
        slab_floats += nadm_decode_slab_floats(d.M, hd.kp[h], p->slices_cap[h]);
    }
    p->p3_slices_cap = nadm_encode_slices_max(d.bmax, d.M, hd.CP);
    bool ok = true;
    auto stream_ok = [&](hipStream_t* s) { ok = ok && hipStreamCreateWithFlags(s, hipStreamNonBlocking) == hipSuccess; };
    auto event_ok = [&](hipEvent_t* e) { ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess; };
    if (hd.n_heads > 1) {
        p->fan_heads = 2;
        stream_ok(&p->head_stream); event_ok(&p->ev_fork); event_ok(&p->ev_join);
    }
    if (dp) {
        stream_ok(&p->side); stream_ok(&p->side_b);
        event_ok(&p->ev_p2); event_ok(&p->ev_a); event_ok(&p->ev_fork1);
        for (int j = 0; j < p->nb; ++j) {
            event_ok(&p->ev_b[j]); event_ok(&p->ev_g[j]);
            if (j > 0) { stream_ok(&p->fan[j]); event_ok(&p->ev_p1[j]); }
        }
    }
    if (!ok) {
        nadm_plan_destroy(p);
        return fail("nadm_plan_create: cannot create the plan's streams and events");
    }
    *out = p;
    return 0;
}

extern "C" void nadm_plan_destroy(nadm_plan_t* p) {
    if (!p) return;
    for (hipStream_t s : {p->side, p->side_b, p->head_stream})
        if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    for (hipStream_t s : p->fan)
        if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    for (hipEvent_t e : {p->ev_p2, p->ev_a, p->ev_fork1, p->ev_fork, p->ev_join})
        if (e) (void)hipEventDestroy(e);
    for (int j = 0; j < NADM_MAX_BUCKETS; ++j)
        for (hipEvent_t e : {p->ev_p1[j], p->ev_b[j], p->ev_g[j]})
            if (e) (void)hipEventDestroy(e);
    for (auto& v : p->trec)
        for (auto& ab : v) { (void)hipEventDestroy(ab.first); (void)hipEventDestroy(ab.second); }
    for (auto& v : p->trec_bkt)
        for (auto& ab : v) { (void)hipEventDestroy(ab.first); (void)hipEventDestroy(ab.second); }
    for (hipEvent_t e : p->pool) (void)hipEventDestroy(e);
    delete p;
}

extern "C" int nadm_plan_set_rows(nadm_plan_t* p, const uint8_t* xp) {
    if (!p || !xp) return fail("nadm_plan_set_rows: null pointer");
    p->d.xp = xp;
    return 0;
}

extern "C" int nadm_plan_set_labels(nadm_plan_t* p, const int32_t* labels, int32_t n_classes, float weight) {
    if (!p) return fail("nadm_plan_set_labels: null pointer");
    if (labels && (p->d.heads.n_heads != 1 || n_classes != p->d.heads.k[0]))
        return fail("nadm_plan_set_labels: supervised mode needs a single head with K == number of classes");     // train.py:79
    p->labels = labels; p->n_classes = n_classes; p->sup_weight = weight;
    return 0;
}

extern "C" int nadm_plan_set_state(nadm_plan_t* p, int32_t step_count, int32_t p_in_unit_range) {
    if (!p || step_count < 0) return fail("nadm_plan_set_state: null pointer or negative step count");
    p->step_count = step_count;
    p->p_unit = p_in_unit_range != 0;
    return 0;
}

extern "C" int32_t nadm_plan_p_in_unit_range(const nadm_plan_t* p) { return p && p->p_unit ? 1 : 0; }

extern "C" int32_t nadm_plan_step_count(const nadm_plan_t* p) { return p ? p->step_count : -1; }

extern "C" int32_t nadm_plan_poisoned(const nadm_plan_t* p) { return p && p->poisoned ? 1 : 0; }

extern "C" int nadm_plan_flush(nadm_plan_t* p, void* stream) {
    if (!p) return fail("nadm_plan_flush: null pointer");
    if (p->poisoned) return poisoned(p, "nadm_plan_flush");
    if (flush_small(p, stream)) return 1;
    // (both stay pending for the next step's first readers: a second wait is free)
    if (p->a_pending) HIP_OK(hipStreamWaitEvent((hipStream_t)stream, p->ev_a, 0), "hipStreamWaitEvent");
    if (p->b_pending)
        for (int j = 0; j < p->nb; ++j) HIP_OK(hipStreamWaitEvent((hipStream_t)stream, p->ev_g[j], 0), "hipStreamWaitEvent");
    return comm_health(p);
}

extern "C" int nadm_plan_infer(nadm_plan_t* p, const int32_t* idx, int32_t b, void* stream) {
    if (!p || !idx) return fail("nadm_plan_infer: null pointer");
    if (b <= 0 || b > p->d.bmax) return fail("nadm_plan_infer: batch size outside (0, bmax]");
    if (!p->d.xp) return fail("nadm_plan_infer: no genotype rows (nadm_plan_set_rows)");
    if (p->poisoned) return poisoned(p, "nadm_plan_infer");
    return forward(p, idx, b, stream);
}

static int step_impl(nadm_plan_t* p, const int32_t* idx, int32_t b, float lr, int32_t with_loss, void* stream) {
    const nadm_plan_desc_t& d = p->d;
    const nadm_heads_t& hd = d.heads;
    const bool dp = d.mode == NADM_MODE_DP, snp = d.mode == NADM_MODE_SNP;
    const float scale = 1.0f / (float)p->world;
    hipStream_t st = (hipStream_t)stream;

    if (forward(p, idx, b, stream)) return 1;
    p->step_count += 1;

    // ---- pass 2: decoder, BCE forward + backward, dP, dQ slab
    if (dp && p->a_pending) {                                    // P of the previous step: final when message A's all-gather is
        HIP_OK(hipStreamWaitEvent(st, p->ev_a, 0), "hipStreamWaitEvent");
        p->a_pending = false;
    }
    Timed t2{p, NADM_T_DECODE_BCE, st};
    if (t2.begin()) return 1;
    const float lr_scale[2] = {lr, dp ? 1.0f : scale};           // (single: scale = 1)
    if (decode_heads(p, idx, b, with_loss, dp ? nullptr : lr_scale, stream)) return 1;
    if (t2.end()) return 1;
    int64_t n_loss = p->n_loss;
    if (p->labels && (!snp || p->rank == 0)) {                   // SNP mode: the supervised term must enter the sum over ranks once
        if (nadm_supervised_ce(d.Q, hd.SP, hd.k[0], hd.kp[0], p->labels, idx, b, p->n_classes, p->sup_weight, d.dqpart, d.losspart + p->n_loss, stream))
            return 1;
        n_loss += 1;
    }
    Timed ta{p, NADM_T_SYNC_A, p->side};
    if (dp) {
        // Message A on the side stream, right behind pass 2: the reduce-scatter and the optimizer launch.  (That launch moves 28 B per
        // element of this rank's slice and costs the step that HBM time wherever it runs -- underneath the MLP backward (here) 11 us at
        // world = 1, underneath pass 3 14, message B 14, the next pass 1 19.5; fewer blocks or a low-priority stream change nothing,
        // profiles/r04_ddp_plan.txt.  At world = W it is 1/W of that.)  Its ALL-GATHER is issued behind message B, at the end of the
        // step: a communicator runs its collectives in the order they were issued, and message B -- which the next pass 1 waits for --
        // must not queue behind 16 MB of P that nothing reads before the next pass 2.
        HIP_OK(hipEventRecord(p->ev_p2, st), "hipEventRecord");
        HIP_OK(hipStreamWaitEvent(p->side, p->ev_p2, 0), "hipStreamWaitEvent");
        if (ta.begin()) return 1;
        if (msg_reduce(p, p->comm_a, p->lay.msg_a_off, p->lay.slice_a, p->side)) return 1;
        if (msg_update(p, p->lay.msg_a_off, p->lay.slice_a, p->lay.slice_b, true, lr, p->side)) return 1;
    }
    p->p_unit = true;                                            // restrict_P ran (epilogue) or runs before P is read next (message A)

    // ---- MLP backward (dQ slab -> dZ, + dZ as pass 3's operand image); the weight gradients are left to pass 3's side blocks
    float* dq_src = d.dqpart;
    int64_t dq_M = d.M;
    if (snp) {
        int64_t o = 0, dq_off = 0;
        for (int h = 0; h < hd.n_heads; ++h) {                   // per head: [chunks, b*kp] -> [b*kp], laid back to back
            if (nadm_sum_rows(d.dqpart + dq_off, p->dec_chunks[h], (int64_t)b * hd.kp[h], d.dqsum + o, stream)) return 1;
            dq_off += p->dec_chunks[h] * (int64_t)b * hd.kp[h];
            o += (int64_t)b * hd.kp[h];
        }
        if (d.comm && d.comm->all_reduce(d.comm->ctx, d.dqsum, (int64_t)b * hd.SP, stream)) return 1;
        dq_src = d.dqsum; dq_M = 1;
    }
    if (hd.CP <= 8 && b < p->last_b && b % 128 != 0) {
        // a shorter batch than the step before: the 32-sample groups between the batch and the end of its last 128-sample tile would
        // keep that step's pieces (pass 3 reads whole tiles; finite pieces meet X = 0 there, a NaN scale would not vanish)
        const int64_t tb = nadm_dz_image_tile_bytes();
        HIP_OK(hipMemsetAsync((char*)d.dzimg + (int64_t)(b / 128) * tb, 0, (size_t)tb, st), "hipMemsetAsync");
    }
    p->last_b = b;
    Timed t3{p, NADM_T_MLP_BWD, st};
    if (t3.begin()) return 1;
    const int64_t nl = with_loss ? n_loss : 0;
    const bool image = hd.CP <= 8;
    const int rc = image ? nadm_mlp_bwd_image(&hd, d.params, dq_src, dq_M, b, d.Z, d.rinv, d.Zn, d.H, d.Q, d.dL, d.dHpre, d.dgp, d.small_part, d.dZ,
                                              nullptr, d.losspart, nl, d.loss_acc, d.dzimg, d.dzcnt, stream)
                         : nadm_mlp_bwd(&hd, d.params, dq_src, dq_M, b, d.Z, d.rinv, d.Zn, d.H, d.Q, d.dL, d.dHpre, d.dgp, d.small_part, d.dZ,
                                        nullptr, d.losspart, nl, d.loss_acc, stream);
    if (rc) return 1;
    if (t3.end()) return 1;

    // ---- pass 3: dV = X^T.dZ from the batch copy pass 2 left (C <= 8) + the MLP weight-gradient partials as side blocks
    const nadm_mlp_weights_t mw{&hd, d.Zn, d.H, d.dL, d.dHpre, d.dgp, d.small_part};
    float* V = d.params + p->lay.off_v;
    float* dV = d.grads + p->lay.off_v;
    const int splits = nadm_sample_splits(b);
    const int p3_flags = image ? NADM_X_CLEAN : 0;
    if (dp) {
        // Message B.  n_buckets > 1, bucket by bucket (DDP's bucketed exchange, neural_admixture.py:315-319): pass 3 is launched range by range; behind
        // range j the side stream sums bucket j over the ranks, updates this rank's slice and gathers the range -- while the compute
        // stream is at range j + 1 -- and the next step's pass 1 consumes the ranges in the same order (encode_fwd_parts).  The MLP
        // weight-gradient partials ride in the FIRST range's launch, so the small parameters travel in bucket 0.  p3_whole: one
        // launch, every bucket starts behind it.  The host queues everything in one go; on every rank the communicator sees
        // RS A, (RS B_j, AG B_j) j = 0.., AG A.
        // ONE bucket (the default): the whole of message B on the COMPUTE stream behind pass 3 -- the next pass 1 needs all of V and
        // nothing is left to overlap with, so a side stream would only add its two cross-stream hand-offs (measured: ~9.5 us each on
        // the GPU's timeline, profiles/r05_rank_emulation.txt).
        const bool parts = p->nb > 1 && !d.p3_whole;
        hipStream_t sb = p->nb > 1 ? p->side_b : st;
        Timed t4{p, NADM_T_ENCODE_BWD, st};
        Timed tb{p, NADM_T_SYNC_B, sb};
        if (t4.begin()) return 1;
        for (int j = 0; j < p->nb; ++j) {
            if (parts || j == 0) {
                const int64_t m0 = parts ? p->lay.bkt_m0[j] : 0, m1 = parts ? p->lay.bkt_m0[j + 1] : d.M;
                const uint8_t* xsrc = image ? d.xg + (m0 / 4) * (int64_t)b : d.xp + m0 / 4;
                if (nadm_encode_bwd_step(xsrc, d.ld, idx, b, m1 - m0, d.dZ, image ? d.dzimg : nullptr, hd.CP, V + m0 * hd.CP, dV + m0 * hd.CP, nullptr,
                                         j == 0 ? &mw : nullptr, p3_flags, stream))
                    return 1;
                if (!parts || j == p->nb - 1) { if (t4.end()) return 1; }
                if (p->nb > 1)
                    for (int i = j; i < (parts ? j + 1 : p->nb); ++i) HIP_OK(hipEventRecord(p->ev_b[i], st), "hipEventRecord");
            }
            if (p->nb > 1) HIP_OK(hipStreamWaitEvent(sb, p->ev_b[j], 0), "hipStreamWaitEvent");
            if (j == 0 && tb.begin()) return 1;
            Timed tj{p, NADM_T_SYNC_B, sb, &p->trec_bkt[j]};
            if (tj.begin()) return 1;
            if (j == 0 && nadm_small_grads(d.small_part, splits, hd.n_small, d.grads, d.params, nullptr, sb)) return 1;     // the sum only
            if (msg_reduce(p, d.comm, p->lay.bkt_off[j], p->lay.bkt_slice[j], sb)) return 1;
            if (msg_update(p, p->lay.bkt_off[j], p->lay.bkt_slice[j], p->lay.bkt_mom[j], false, lr, sb)) return 1;
            if (msg_gather(p, d.comm, p->lay.bkt_off[j], p->lay.bkt_slice[j], sb)) return 1;
            if (tj.end()) return 1;
            if (p->nb > 1) HIP_OK(hipEventRecord(p->ev_g[j], sb), "hipEventRecord");
        }
        if (tb.end()) return 1;
        p->b_pending = p->nb > 1;
        if (msg_gather(p, p->comm_a, p->lay.msg_a_off, p->lay.slice_a, p->side)) return 1;     // message A's all-gather: behind B in the communicator's order
        if (ta.end()) return 1;
        HIP_OK(hipEventRecord(p->ev_a, p->side), "hipEventRecord");
        p->a_pending = true;
        return d.debug ? comm_health(p) : 0;
    }
    Timed t4{p, NADM_T_ENCODE_BWD, st};
    if (t4.begin()) return 1;
    const nadm_adam_t av = adam_at(p, p->lay.off_v, lr, p->step_count, scale);
    const int p3_slices = (d.p3_slab && image) ? nadm_encode_slices(b, d.M, hd.CP) : 1;      // sample slices where the SNP chunks alone leave CUs idle
    if (p3_slices > p->p3_slices_cap) return fail("nadm_step: pass 3 would be cut into more sample slices than the plan's slab was sized for at creation");
    if (p3_slices > 1) {
        if (nadm_encode_bwd_sliced(d.xg, d.ld, idx, b, d.M, d.dZ, d.dzimg, hd.CP, V, dV, &av, &mw, p3_flags, p3_slices, d.p3_slab, d.p3_cnt, stream)) return 1;
    } else if (nadm_encode_bwd_step(image ? d.xg : d.xp, d.ld, idx, b, d.M, d.dZ, image ? d.dzimg : nullptr, hd.CP, V, dV, &av, &mw, p3_flags, stream)) return 1;
    if (t4.end()) return 1;

    // ---- the small parameters
    if (hd.CP <= 8) {                                            // rides in the next pass 1
        p->small_pending = true;
        p->pend_splits = splits; p->pend_lr = lr; p->pend_scale = scale; p->pend_step = p->step_count;
        return (d.debug && snp) ? comm_health(p) : 0;
    }
    const nadm_adam_t sa = adam_at(p, 0, lr, p->step_count, scale);
    if (nadm_small_grads(d.small_part, splits, hd.n_small, d.grads, d.params, &sa, stream)) return 1;
    return (d.debug && snp) ? comm_health(p) : 0;
}

extern "C" int nadm_step(nadm_plan_t* p, const int32_t* idx, int32_t b, float lr, int32_t with_loss, void* stream) {
    if (!p || !idx) return fail("nadm_step: null pointer");
    if (b <= 0 || b > p->d.bmax) return fail("nadm_step: batch size outside (0, bmax]");
    if (!p->d.xp) return fail("nadm_step: no genotype rows (nadm_plan_set_rows)");
    if (p->poisoned) return poisoned(p, "nadm_step");
    // (up to here nothing was touched; from here on a failure leaves step count, hand-offs and side streams half-way)
    const int rc = step_impl(p, idx, b, lr, with_loss, stream);
    if (rc) p->poisoned = true;
    return rc;
}

extern "C" int nadm_plan_timing(nadm_plan_t* p, uint32_t mask) {
    if (!p) return fail("nadm_plan_timing: null pointer");
    p->tmask = mask & ((1u << NADM_T_COUNT) - 1);
    return 0;
}

static int mean_ms(nadm_plan* p, std::vector<std::pair<hipEvent_t, hipEvent_t>>& rec, float* ms, int32_t* count) {
    double sum = 0.0;
    for (auto& ab : rec) {
        float t = 0.f;
        HIP_OK(hipEventElapsedTime(&t, ab.first, ab.second), "hipEventElapsedTime");
        sum += t;
        p->pool.push_back(ab.first);
        p->pool.push_back(ab.second);
    }
    *ms = rec.empty() ? 0.f : (float)(sum / (double)rec.size());
    if (count) *count = (int32_t)rec.size();
    rec.clear();
    return 0;
}

extern "C" int nadm_plan_kernel_ms(nadm_plan_t* p, float* ms, int32_t* counts) {
    if (!p || !ms) return fail("nadm_plan_kernel_ms: null pointer");
    HIP_OK(hipDeviceSynchronize(), "hipDeviceSynchronize");
    for (int i = 0; i < NADM_T_COUNT; ++i)
        if (mean_ms(p, p->trec[i], &ms[i], counts ? &counts[i] : nullptr)) return 3;
    float dummy;
    for (auto& v : p->trec_bkt)
        if (mean_ms(p, v, &dummy, nullptr)) return 3;
    return 0;
}

extern "C" int nadm_plan_bucket_ms(nadm_plan_t* p, float* ms, int32_t* n) {
    if (!p || !ms || !n) return fail("nadm_plan_bucket_ms: null pointer");
    HIP_OK(hipDeviceSynchronize(), "hipDeviceSynchronize");
    *n = p->d.mode == NADM_MODE_DP ? p->nb : 0;
    for (int j = 0; j < *n; ++j)
        if (mean_ms(p, p->trec_bkt[j], &ms[j], nullptr)) return 3;
    return 0;
}
