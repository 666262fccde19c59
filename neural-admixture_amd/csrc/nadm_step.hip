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
// Everything else (P / V in single and SNP mode, small | V in DP mode) is final when the step's last launch on `stream` is.
#include "../../include/nadm.h"
#include "nadm_host.h"
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and prototypes only: the functions are resolved with dlsym (no link-time dependency)
#include <stdlib.h>
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

extern "C" int nadm_flat_layout(const nadm_heads_t* hd, int64_t M, int32_t world, nadm_flat_layout_t* out) {
    if (!hd || !out) return fail("nadm_flat_layout: null pointer");
    if (M <= 0 || world < 1 || hd->n_heads < 1 || hd->n_heads > NADM_MAX_HEADS) return fail("nadm_flat_layout: M, world >= 1 and 1..32 heads");
    memset(out, 0, sizeof(*out));
    out->off_v = round_up(hd->n_small, 64);
    const int64_t b_len = out->off_v + M * hd->CP;
    out->slice_b = round_up((b_len + world - 1) / world, 4);            // 16-byte slices: what the Adam kernel's vector accesses need
    out->msg_a_off = out->slice_b * world;
    int64_t off = out->msg_a_off;
    for (int h = 0; h < hd->n_heads; ++h) {
        out->off_p[h] = off;
        off += M * hd->kp[h];
    }
    out->slice_a = round_up((off - out->msg_a_off + world - 1) / world, 4);
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
void rccl_destroy(void* ctx) {
    RcclCtx* c = static_cast<RcclCtx*>(ctx);
    if (c->comm) c->f.CommDestroy(c->comm);
    delete c;                                                   // (the library stays loaded: other communicators may use it)
}

int noop_slices(void*, float*, int64_t, void*) { return 0; }

}  // namespace

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

extern "C" int nadm_comm_rccl(const char* librccl_path, const void* id128, int32_t rank, int32_t world, nadm_comm_t** out) {
    if (!id128 || !out) return fail("nadm_comm_rccl: null pointer");
    if (world < 1 || rank < 0 || rank >= world) return fail("nadm_comm_rccl: need 0 <= rank < world");
    RcclCtx* c = new RcclCtx;
    if (rccl_load(librccl_path, &c->f)) { delete c; return 1; }
    c->rank = rank; c->world = world;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    const ncclResult_t r = c->f.CommInitRank(&c->comm, world, id, rank);          // blocks until every rank has called it
    if (r != ncclSuccess) {
        const int rc = rccl_fail(c, "ncclCommInitRank", r);
        delete c;
        return rc;
    }
    nadm_comm_t* k = new nadm_comm_t;
    k->rank = rank; k->world = world; k->ctx = c;
    k->reduce_scatter = rccl_reduce_scatter; k->all_gather = rccl_all_gather; k->all_reduce = rccl_all_reduce; k->destroy = rccl_destroy;
    *out = k;
    return 0;
}

extern "C" int nadm_comm_emulated(int32_t world, nadm_comm_t** out) {
    if (!out || world < 1) return fail("nadm_comm_emulated: world >= 1");
    nadm_comm_t* k = new nadm_comm_t;
    k->rank = 0; k->world = world; k->ctx = nullptr;
    k->reduce_scatter = noop_slices; k->all_gather = noop_slices; k->all_reduce = noop_slices; k->destroy = nullptr;
    *out = k;
    return 0;
}

extern "C" void nadm_comm_free(nadm_comm_t* comm) {
    if (!comm) return;
    if (comm->destroy) comm->destroy(comm->ctx);
    delete comm;
}

// ------------------------------------------------------------------------------------------------- the plan
struct nadm_plan {
    nadm_plan_desc_t d;
    nadm_flat_layout_t lay;
    int rank = 0, world = 1;
    int step_count = 0;
    bool p_unit = true;
    // what the last step left to the next one (see the head of this file)
    bool small_pending = false;
    int pend_splits = 0, pend_step = 0;
    float pend_lr = 0.f, pend_scale = 1.f;
    bool a_pending = false;
    int last_b = 0;                                              // batch size of the previous step (dZ image hygiene, see nadm_step)
    hipStream_t side = nullptr;                                  // DP: message A
    hipEvent_t ev_p2 = nullptr, ev_a = nullptr;
    // multi-head models: the heads' pass-2 launches are independent (own P rows, slab, loss slots; they share X and Q) and each ends
    // in a partly filled round of blocks: two launches in flight fill each other's tails (r02: 3.12 -> 2.9 ms at nine heads)
    int fan = 1;
    hipStream_t head_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    const int32_t* labels = nullptr;
    int n_classes = 0;
    float sup_weight = 0.f;
    int64_t enc_chunks = 0, dec_chunks[NADM_MAX_HEADS] = {0}, loss_off[NADM_MAX_HEADS] = {0}, n_loss = 0;
    uint32_t tmask = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> trec[NADM_T_COUNT];
    std::vector<hipEvent_t> pool;
};

namespace {

struct Timed {                                                   // RAII-free: begin() / end() around a group of launches on one stream
    nadm_plan* p;
    int slot;
    hipStream_t st;
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
    int end() {
        if (!a) return 0;
        HIP_OK(hipEventRecord(b, st), "hipEventRecord");
        p->trec[slot].emplace_back(a, b);
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

// pass 1 (+ the small update the previous step left to it) and the MLP forward; SNP mode: the partial Z summed over ranks in between
int forward(nadm_plan* p, const int32_t* idx, int b, void* stream) {
    const nadm_plan_desc_t& d = p->d;
    const nadm_heads_t& hd = d.heads;
    float* V = d.params + p->lay.off_v;
    Timed t1{p, NADM_T_ENCODE_FWD, (hipStream_t)stream};
    if (t1.begin()) return 1;
    if (p->small_pending && hd.CP <= 8) {
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
    const int fan = hd.n_heads > 1 ? p->fan : 1;
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
        if (d.qimg && kp <= 16)
            rc = nadm_decode_bce_images(d.xp, d.ld, idx, b, d.M, Ph, kp, Qh, hd.SP, dPh, slab, lossp, flags, xg, adp,
                                        (const char*)d.qimg + (int64_t)h * d.qimg_head_bytes, st);
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

// one message of the sample-sharded step = reduce-scatter of the gradients -> Adam (+ restrict_P) on this rank's slice -> all-gather of
// the updated parameters, as three calls: message A issues them at different points of the step (nadm_step).  msg_off / slice in
// floats of the flat buffers; mom_off = where the slice's moments start in d.m / d.v
int msg_reduce(nadm_plan* p, int64_t msg_off, int64_t slice, void* stream) {
    const nadm_comm_t* c = p->d.comm;
    return (c && c->reduce_scatter(c->ctx, p->d.grads + msg_off, slice, stream)) ? 1 : 0;
}
int msg_update(nadm_plan* p, int64_t msg_off, int64_t slice, int64_t mom_off, bool clamp, float lr, void* stream) {
    const nadm_plan_desc_t& d = p->d;
    const int64_t lo = msg_off + (int64_t)p->rank * slice;
    return nadm_adam(d.params + lo, d.grads + lo, d.m + mom_off, d.v + mom_off, slice, clamp ? 0 : slice, lr, p->step_count, 1.0f / (float)p->world, stream);
}
int msg_gather(nadm_plan* p, int64_t msg_off, int64_t slice, void* stream) {
    const nadm_comm_t* c = p->d.comm;
    return (c && c->all_gather(c->ctx, p->d.params + msg_off, slice, stream)) ? 1 : 0;
}

}  // namespace

extern "C" int nadm_plan_create(const nadm_plan_desc_t* desc, nadm_plan_t** out) {
    if (!desc || !out) return fail("nadm_plan_create: null pointer");
    const nadm_plan_desc_t& d = *desc;
    if (d.mode != NADM_MODE_SINGLE && d.mode != NADM_MODE_DP && d.mode != NADM_MODE_SNP) return fail("nadm_plan_create: unknown mode");
    if (d.M <= 0 || d.bmax <= 0 || d.ld % 16 != 0 || d.ld * 4 < d.M) return fail("nadm_plan_create: M, bmax > 0; ld a multiple of 16 and >= ceil(M/4)");
    const nadm_heads_t& hd = d.heads;
    if (hd.n_heads < 1 || hd.n_heads > NADM_MAX_HEADS) return fail("nadm_plan_create: head table not initialised (nadm_heads_init)");
    if (!d.params || !d.grads || !d.m || !d.v || !d.zpart || !d.Z || !d.rinv || !d.Zn || !d.H || !d.Q || !d.dL || !d.dHpre || !d.dgp || !d.dZ ||
        !d.dqpart || !d.losspart || !d.small_part || !d.loss_acc)
        return fail("nadm_plan_create: null pointer in the descriptor");
    if (hd.CP <= 8 && (!d.dzimg || !d.dzcnt || !d.xg)) return fail("nadm_plan_create: C <= 8 needs dzimg, dzcnt and xg");
    if (d.mode == NADM_MODE_SNP && (!d.zsum || !d.dqsum)) return fail("nadm_plan_create: SNP mode needs zsum and dqsum");
    if (d.mode == NADM_MODE_SINGLE && d.comm && d.comm->world != 1) return fail("nadm_plan_create: single mode with a communicator of several ranks");
    if (d.comm && (!d.comm->reduce_scatter || !d.comm->all_gather || !d.comm->all_reduce || d.comm->world < 1 || d.comm->rank < 0 ||
                   d.comm->rank >= d.comm->world))
        return fail("nadm_plan_create: incomplete communicator");
    nadm_plan* p = new nadm_plan;
    p->d = d;
    p->world = d.comm ? d.comm->world : 1;
    p->rank = d.comm ? d.comm->rank : 0;
    if (nadm_flat_layout(&hd, d.M, d.mode == NADM_MODE_DP ? p->world : 1, &p->lay)) { delete p; return 1; }
    p->enc_chunks = nadm_encode_chunks(d.M);
    for (int h = 0; h < hd.n_heads; ++h) {
        p->dec_chunks[h] = nadm_decode_chunks(d.M, hd.kp[h]);
        p->loss_off[h] = p->n_loss;
        p->n_loss += p->dec_chunks[h];
    }
    if (hd.n_heads > 1) {
        p->fan = 2;
        if (hipStreamCreateWithFlags(&p->head_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming) != hipSuccess) {
            nadm_plan_destroy(p);
            return fail("nadm_plan_create: cannot create the second pass-2 stream");
        }
    }
    if (d.mode == NADM_MODE_DP) {
        if (hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&p->ev_p2, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&p->ev_a, hipEventDisableTiming) != hipSuccess) {
            nadm_plan_destroy(p);
            return fail("nadm_plan_create: cannot create the side stream of message A");
        }
    }
    *out = p;
    return 0;
}

extern "C" void nadm_plan_destroy(nadm_plan_t* p) {
    if (!p) return;
    if (p->side) { (void)hipStreamSynchronize(p->side); (void)hipStreamDestroy(p->side); }
    if (p->head_stream) { (void)hipStreamSynchronize(p->head_stream); (void)hipStreamDestroy(p->head_stream); }
    for (hipEvent_t e : {p->ev_p2, p->ev_a, p->ev_fork, p->ev_join})
        if (e) (void)hipEventDestroy(e);
    for (auto& v : p->trec)
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

extern "C" int nadm_plan_flush(nadm_plan_t* p, void* stream) {
    if (!p) return fail("nadm_plan_flush: null pointer");
    if (flush_small(p, stream)) return 1;
    if (p->a_pending) HIP_OK(hipStreamWaitEvent((hipStream_t)stream, p->ev_a, 0), "hipStreamWaitEvent");    // (stays pending for the next pass 2: a second wait is free)
    return 0;
}

extern "C" int nadm_plan_infer(nadm_plan_t* p, const int32_t* idx, int32_t b, void* stream) {
    if (!p || !idx) return fail("nadm_plan_infer: null pointer");
    if (b <= 0 || b > p->d.bmax) return fail("nadm_plan_infer: batch size outside (0, bmax]");
    if (!p->d.xp) return fail("nadm_plan_infer: no genotype rows (nadm_plan_set_rows)");
    return forward(p, idx, b, stream);
}

extern "C" int nadm_step(nadm_plan_t* p, const int32_t* idx, int32_t b, float lr, int32_t with_loss, void* stream) {
    if (!p || !idx) return fail("nadm_step: null pointer");
    if (b <= 0 || b > p->d.bmax) return fail("nadm_step: batch size outside (0, bmax]");
    if (!p->d.xp) return fail("nadm_step: no genotype rows (nadm_plan_set_rows)");
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
        if (msg_reduce(p, p->lay.msg_a_off, p->lay.slice_a, p->side)) return 1;
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
    Timed t4{p, NADM_T_ENCODE_BWD, st};
    if (t4.begin()) return 1;
    const nadm_mlp_weights_t mw{&hd, d.Zn, d.H, d.dL, d.dHpre, d.dgp, d.small_part};
    const nadm_adam_t av = adam_at(p, p->lay.off_v, lr, p->step_count, scale);
    float* V = d.params + p->lay.off_v;
    float* dV = d.grads + p->lay.off_v;
    if (nadm_encode_bwd_step(image ? d.xg : d.xp, d.ld, idx, b, d.M, d.dZ, image ? d.dzimg : nullptr, hd.CP, V, dV, dp ? nullptr : &av, &mw,
                             image ? NADM_X_CLEAN : 0, stream))
        return 1;
    if (t4.end()) return 1;

    // ---- the small parameters, and in DP mode message B
    const int splits = nadm_sample_splits(b);
    if (dp) {
        Timed tb{p, NADM_T_SYNC_B, st};
        if (tb.begin()) return 1;
        if (nadm_small_grads(d.small_part, splits, hd.n_small, d.grads, d.params, nullptr, stream)) return 1;     // the sum only
        if (msg_reduce(p, 0, p->lay.slice_b, stream)) return 1;
        if (msg_update(p, 0, p->lay.slice_b, 0, false, lr, stream)) return 1;
        if (msg_gather(p, 0, p->lay.slice_b, stream)) return 1;
        if (tb.end()) return 1;
        if (msg_gather(p, p->lay.msg_a_off, p->lay.slice_a, p->side)) return 1;     // message A's all-gather: behind B in the communicator's order
        if (ta.end()) return 1;
        HIP_OK(hipEventRecord(p->ev_a, p->side), "hipEventRecord");
        p->a_pending = true;
        return 0;
    }
    if (hd.CP <= 8) {                                            // rides in the next pass 1
        p->small_pending = true;
        p->pend_splits = splits; p->pend_lr = lr; p->pend_scale = scale; p->pend_step = p->step_count;
        return 0;
    }
    const nadm_adam_t sa = adam_at(p, 0, lr, p->step_count, scale);
    return nadm_small_grads(d.small_part, splits, hd.n_small, d.grads, d.params, &sa, stream);
}

extern "C" int nadm_plan_timing(nadm_plan_t* p, uint32_t mask) {
    if (!p) return fail("nadm_plan_timing: null pointer");
    p->tmask = mask & ((1u << NADM_T_COUNT) - 1);
    return 0;
}

extern "C" int nadm_plan_kernel_ms(nadm_plan_t* p, float* ms, int32_t* counts) {
    if (!p || !ms) return fail("nadm_plan_kernel_ms: null pointer");
    HIP_OK(hipDeviceSynchronize(), "hipDeviceSynchronize");
    for (int i = 0; i < NADM_T_COUNT; ++i) {
        double sum = 0.0;
        for (auto& ab : p->trec[i]) {
            float t = 0.f;
            HIP_OK(hipEventElapsedTime(&t, ab.first, ab.second), "hipEventElapsedTime");
            sum += t;
            p->pool.push_back(ab.first);
            p->pool.push_back(ab.second);
        }
        ms[i] = p->trec[i].empty() ? 0.f : (float)(sum / (double)p->trec[i].size());
        if (counts) counts[i] = (int32_t)p->trec[i].size();
        p->trec[i].clear();
    }
    return 0;
}
