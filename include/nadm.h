/*
 * nadm.h -- C ABI of the MI355X-native Neural ADMIXTURE training engine (libnadm.so).
 *
 * This is the drop-in boundary for the reference's hot path.  Every entry point names the
 * reference interface it replaces (paths relative to the reference repo).  Conventions:
 *   - plain pointers and sizes only; no torch / ATen types.
 *   - every device pointer is CALLER-OWNED (the reference's contract too: the caller allocates
 *     both tensors of pack2bit_cpu_to_gpu / unpack2bit_gpu_to_gpu, train.py:121,
 *     neural_admixture.py:377,405); the library never allocates or frees device memory.
 *   - every call is ASYNCHRONOUS on the hipStream_t passed as `stream` (void*; 0 = null stream)
 *     unless stated otherwise.  (The reference's extension blocks with cudaDeviceSynchronize on the
 *     legacy default stream, pack2bit.cu:115,141; callers that want that behaviour synchronise.)
 *   - return value: 0 = ok, nonzero = error; nadm_last_error() returns a thread-local message
 *     (the reference raises RuntimeError through TORCH_CHECK, pack2bit.cu:67-76,121-130; the
 *     Python host mirror turns nonzero statuses into RuntimeError).
 *   - one host thread per process/GPU, not re-entrant per stream (same as the reference).
 *
 * Device data layout (all float32 unless noted):
 *   xp     uint8 [rows, ld]   2-bit packed genotypes, sample-major, SNP 4c+i in bits [2i,2i+1] of
 *                             byte c (pack2bit.cu:26-31); ld >= ceil(M/4), ld % 16 == 0, ld < 2^32 (the matrix-pipe passes form row addresses from
 *                             32-bit factors and refuse longer rows), pad bytes 0.
 *   V      [M, CP]            encoder projection, CP = C rounded up to a multiple of 4, pad cols 0
 *                             (reference: Q_P.V [M,C], neural_admixture.py:129-130).
 *   P_h    [M, KP_h]          decoder head h, SNP-major, KP_h = padded K (nadm_pad_k), pad cols 0
 *                             (reference: decoders[h].weight, logical [M,k], neural_admixture.py:73-74).
 *   small  flat               g[C] | W1[Hd,C] | b1[Hd] | for h: Wk_h[k_h,Hd] | bk_h[k_h]
 *                             (batch_norm.weight, common_encoder.0.{weight,bias},
 *                              multihead_encoder.heads.h.{weight,bias}).
 *   Q      [b, SP]            per-head softmax outputs; head h occupies columns
 *                             [qoff_h, qoff_h + k_h), SP = sum KP_h; pad cols 0.
 */
#ifndef NADM_H
#define NADM_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NADM_MAX_HEADS 32
#define NADM_MAX_K 64
#define NADM_MAX_BUCKETS 8
#define NADM_MAX_P2_SLICES 8   /* sample slices of pass 2 (nadm_decode_bce_sliced) */
#define NADM_ABI_VERSION 14  /* 14: nadm_clock_probe; 13: pass 3 in sample slices (nadm_encode_bwd_sliced, nadm_encode_slices(_max), nadm_encode_slab_floats, nadm_encode_bwd_chunks, nadm_plan_desc_t.p3_slab / p3_cnt); 12: nadm_test_force_slices / nadm_test_force_generic_mlp exist in the test build only (-DNADM_TEST_HOOKS), nadm_calib_clock / nadm_wall_clock_khz; 11: nadm_gmm_fit_means_dev, nadm_loglik_blocks counts 8 row slices per 1024-SNP block, nadm_decode_bce_sliced / nadm_decode_slices / nadm_decode_slab_floats / nadm_test_force_slices + nadm_plan_desc_t.p2_slab / p2_cnt (pass 2 in sample slices when the SNP chunks alone do not fill the chip); 10: message B of the sample-sharded step in SNP-range buckets (nadm_flat_layout takes n_buckets, nadm_flat_layout_t.bkt_*, nadm_plan_desc_t.n_buckets / p3_whole / comm_a / debug, nadm_encode_fwd_part, nadm_plan_bucket_ms), nadm_comm_t.async_error, nadm_comm_rccl_probe, nadm_comm_rccl with a watchdog (timeout_ms), a failed step poisons its plan; 9: nadm_step / nadm_plan_* / nadm_comm_* / nadm_flat_layout (the step as one call, sharded optimizer), nadm_test_force_generic_mlp; 8: nadm_dz_image(_bytes), nadm_mlp_bwd_image; nadm_encode_bwd, nadm_encode_bwd_step, nadm_pca_project_t take the operand image of dZ / Y; 7: nadm_encode_fwd_step, nadm_sum_rows, dqpart of nadm_mlp_bwd is float* (folded in place); 6: nadm_mlp_fwd_images, nadm_decode_bce_images, nadm_q_image_bytes, nadm_encode_fwd_small; 5: nadm_adam_t.when, nadm_adam2, with_loss bit 1; 4: nadm_decode_bce_step, nadm_encode_bwd_step (nadm_adam_t, nadm_mlp_weights_t), nadm_small_grads; 3: nadm_decode_bce_gather; 2: nadm_mlp_bwd_weights, nadm_supervised_ce, nadm_pca_project(_t), nadm_loglik, nadm_savetxt_f32, nadm_decode_chunk_snps; grad_small of nadm_mlp_bwd may be NULL */

/* Head table shared by the MLP entry points (mirror of NeuralEncoder/NeuralDecoder's ks list,
 * neural_admixture.py:27-29,66-76). Offsets are element offsets into the `small` flat buffer. */
typedef struct nadm_heads {
    int32_t n_heads;
    int32_t C;              /* n_components (true) */
    int32_t CP;             /* padded */
    int32_t Hd;             /* hidden size */
    int32_t SP;             /* sum of padded K = row length of Q / dQ */
    int32_t n_small;        /* number of floats in the small flat buffer */
    int32_t k[NADM_MAX_HEADS];
    int32_t kp[NADM_MAX_HEADS];
    int32_t qoff[NADM_MAX_HEADS];
    int32_t wk_off[NADM_MAX_HEADS];
    int32_t bk_off[NADM_MAX_HEADS];
    int32_t g_off, w1_off, b1_off;
} nadm_heads_t;

/* ---- introspection ------------------------------------------------------------------- */
int         nadm_abi_version(void);
const char* nadm_last_error(void);
/* Padded widths used by the kernels: multiple of 4 up to 16, then {24,32,48,64}. <=0 if unsupported. */
int         nadm_pad_k(int k);
/* Fill a head table from (C, Hd, ks[n]) -- the layout every other call assumes. */
int         nadm_heads_init(nadm_heads_t* out, int C, int Hd, const int32_t* ks, int n);
/* Number of SNP chunks (= rows of partial buffers) the three genotype passes use for M SNPs. */
int64_t     nadm_encode_chunks(int64_t M);      /* zpart  is [chunks, b, CP]  */
int64_t     nadm_decode_chunks(int64_t M, int kp); /* per head: dqpart slab [chunks, b, kp], losspart [chunks] */
/* SNPs per chunk of the decoder pass.  nadm_decode_bce / nadm_encode_bwd may be launched on an SNP sub-range
 * [m0, m1) with m0 a multiple of lcm(this, 1024): pass xp + m0/4, P/dP (V/dV) + m0*kp, M = m1 - m0 and the slab /
 * loss pointers advanced by m0/chunk_snps rows (a caller that wants a piece of a gradient early). */
int32_t     nadm_decode_chunk_snps(int kp);
int32_t     nadm_sample_splits(int b);          /* small_part is [splits, n_small] */

/* ---- a1/a2: packing  (replaces pack2bit.cu:10-36,65-117 and :38-62,120-142) ------------ */
/* Host-side pack: g_host uint8 [N,M] (row stride M) -> out_host [N,ld].  Synchronous, OpenMP. */
int nadm_pack2bit_host(const uint8_t* g_host, uint8_t* out_host, int64_t N, int64_t M, int64_t ld);
/* Device pack kernel: g_dev uint8 [rows,M] unpacked on device -> out_dev [rows,ld] (pad bytes zeroed). */
int nadm_pack2bit(const uint8_t* g_dev, uint8_t* out_dev, int64_t rows, int64_t M, int64_t ld, void* stream);
/* Device unpack (test / interop only; the training kernels decode in registers):
 * in_dev [rows,ld] -> out_dev uint8 [rows,M]. */
int nadm_unpack2bit(const uint8_t* in_dev, uint8_t* out_dev, int64_t rows, int64_t M, int64_t ld, void* stream);

/* ---- 8(f)-1: PLINK .bed -> packed sample-major, no uint8 [N,M] detour -------------------------
 * (replaces SNPReader._read_bed + utils_c.read_bed + pack2bit for BED input: src/snp_reader.py:16-45,
 * src/utils_c/utils.pyx:43-67, pack2bit.cu:65-117).  bed = the file contents AFTER the 3 magic bytes,
 * SNP-major [M, ceil(N/4)], 4 samples per byte, PLINK codes mapped with the reference's table [2,3,1,0]
 * (0b01 = missing -> 3).  out_host [N, ld] gets the layout every kernel here consumes.  counts[4] receives
 * the number of genotypes with code 0..3 (before any flip).  If flip_if_mean_ge1 != 0 and the mean code
 * (3s included, like the reference's G.mean(), snp_reader.py:110) is >= 1, alleles are flipped 0<->2 and
 * *flipped is set to 1; missing stays 3 (the reference's uint8 `2 - G` turns 3 into 255, which its own GPU
 * path masks back to 3, pack2bit.cu:29).  Synchronous, multi-threaded. */
int nadm_bed_to_packed(const uint8_t* bed, int64_t N, int64_t M, uint8_t* out_host, int64_t ld,
                       int64_t* counts, int32_t flip_if_mean_ge1, int32_t* flipped);

/* The same conversion on the device: bed_dev = the file contents after the 3 magic bytes, already in HBM ([M, ceil(N/4)]);
 * out_dev [N, ld] (ld % 16 == 0; the row padding is written as zeros); counts_dev uint64[4] and flipped_dev int32[1] are
 * device scratch/outputs (zeroed here).  The flip decision is taken on the device, nothing is read back: asynchronous. */
int nadm_bed_to_packed_dev(const uint8_t* bed_dev, int64_t N, int64_t M, uint8_t* out_dev, int64_t ld, uint64_t* counts_dev,
                           int32_t flip_if_mean_ge1, int32_t* flipped_dev, void* stream);

/* ---- a4/a5: encoder projection  Z = X.V  (neural_admixture.py:169-172) ----------------- */
/* rows idx[0..b) of xp are the batch (replaces Dataset_admixture.__getitem__ + collate,
 * loaders.py:62-72, and the per-step unpack2bit_gpu_to_gpu, neural_admixture.py:404-406).
 * Writes per-chunk partial sums zpart [nadm_encode_chunks(M), b, CP]; nadm_mlp_fwd reduces them. */
int nadm_encode_fwd(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                    const float* V, int32_t CP, float* zpart, void* stream);

/* nadm_encode_fwd on an SNP sub-range that is ONE PART of a pass launched in several (the sample-sharded step: one part per
 * bucket of message B, each part waiting for its own bucket, include "The training step as ONE call" below): pointers already
 * advanced to the range -- xp + m0/4, V + m0*CP, zpart + (m0/2048)*b*CP, M = m1 - m0, m0 a multiple of 2048 -- and total_chunks =
 * nadm_encode_chunks of the WHOLE pass, which the batch split is chosen for (the parts share the device; a part sized on its own
 * would cut the batch finer and pay the per-block operand build that often).  Same results as the one launch, bit for bit. */
int nadm_encode_fwd_part(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                         const float* V, int32_t CP, float* zpart, int64_t total_chunks, void* stream);

/* ---- 8(f)-3: init-time PCA projection  X_pca = (G/2).V  with a MISSING call counted as 1.5 --------------
 * (train.py:49-55: `batch.astype(np.float32)/2 @ V.T` on the raw codes, no masking; feeds the sklearn GMM.  Also the
 * first tall-skinny product of the randomized SVD, src/svd.py:52,62 -> utils_c/rsvd.pyx multiply_A_omega.)
 * Same kernel, buffers and partial-sum layout as nadm_encode_fwd; CP <= 8 only. */
int nadm_pca_project(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                     const float* V, int32_t CP, float* zpart, void* stream);
/* Transposed product with the same convention,  out [M,CP] = (G/2)^T . Y  for the rows idx[0..b), Y [b,CP]: the second
 * tall-skinny product of the randomized SVD (src/svd.py:60,77 -> utils_c/rsvd.pyx multiply_QT_A; A = raw codes = 2 * G/2).
 * Same kernel as nadm_encode_bwd; CP <= 8 only; yimg = nadm_dz_image of Y. */
int nadm_pca_project_t(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                       const float* Y, const void* yimg, int32_t CP, float* out, void* stream);

/* ---- a6-a8: RMSNorm + Linear/ReLU + per-head Linear + softmax (neural_admixture.py:173-176) */
int nadm_mlp_fwd(const nadm_heads_t* hd, const float* small, const float* zpart, int64_t n_chunks, int32_t b,
                 float* Z, float* rinv, float* Zn, float* H, float* Q, void* stream);
/* The same, and Q additionally as the bf16 MFMA operand images pass 2 builds from it (heads with padded K <= 16): every block
 * of nadm_decode_bce* otherwise splits the batch's Q into bf16 pieces itself, tile by tile (1954 blocks at M = 500k doing the
 * same 800 x K conversion).  qimg: n_heads regions of qimg_head_bytes >= nadm_q_image_bytes(b) bytes, 16-byte aligned and
 * ZERO-FILLED ONCE by the caller (slots that hold no piece are never written); head h's images start at h * qimg_head_bytes.
 * Hand a head's region to nadm_decode_bce_images together with the same Q. */
int64_t nadm_q_image_bytes(int32_t b);
int nadm_mlp_fwd_images(const nadm_heads_t* hd, const float* small, const float* zpart, int64_t n_chunks, int32_t b,
                        float* Z, float* rinv, float* Zn, float* H, float* Q, void* qimg, int64_t qimg_head_bytes, void* stream);

/* ---- a9-a11: decoder Q.P^T -> clamp -> BCE(sum) forward + backward, one head -------------
 * (neural_admixture.py:94-97, :288/:431, autograd of both).  P,dP [M,kp]; Q = Qbase + qoff
 * with row stride SP; dqpart = this head's slab [nadm_decode_chunks(M,kp), b, kp]; losspart
 * [chunks] (written only if with_loss != 0).  Missing genotypes are x = 0 in input and target.
 * with_loss is a bit set: 1 = compute the loss value; 2 = P may hold values outside [0, 1] (before the first restrict_P: the
 * reference's supervised run starts from per-class means of the raw codes, train.py:82) -- the matrix-core kernels then
 * clamp the reconstruction before the logarithm, which they otherwise skip because clamp(r) == r up to rounding. */
int nadm_decode_bce(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                    const float* P, int32_t kp, const float* Q, int32_t SP,
                    float* dP, float* dqpart, float* losspart, int32_t with_loss, void* stream);

/* The same pass with a by-product for pass 3: the batch as a copy of its own in xg -- rows in batch order, every missing call
 * (code 3) already replaced by 0 (the model's input, neural_admixture.py:170), TILED by pass 3's chunks: byte column c of batch
 * row i at  (c / 128) * b * 128 + i * 128 + c % 128  (nadm_batch_copy_bytes(b, M) bytes; an SNP sub-range launch starting at
 * SNP m0 passes xg + (m0 / 4) * b).  Pass 3 (nadm_encode_bwd) is then given xg instead of xp with flags = NADM_X_CLEAN (idx is
 * not read): each of its blocks streams one contiguous b x 128 byte region instead of gathering 128-byte row pieces 125 KB
 * apart out of the resident matrix (those arrive at 2.6 TB/s, and at 12.5 GB resident 13 % of them miss the translation
 * cache).  No reference counterpart: the reference re-gathers the unpacked batch per pass (utils.pyx:43-67). */
int nadm_decode_bce_gather(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                           const float* P, int32_t kp, const float* Q, int32_t SP,
                           float* dP, float* dqpart, float* losspart, int32_t with_loss, uint8_t* xg, void* stream);

/* ---- single-GPU step: Adam applied where the gradient is completed -----------------------------------------------------
 * In a step on one GPU the gradient of a P row is final when the pass-2 block that owns it finishes, and nothing else in
 * the step reads that row again; likewise for V rows and pass 3.  The *_step entry points take the Adam state of those
 * rows and apply optimizer.step() + restrict_P (neural_admixture.py:187-204,411-412) to them in the kernel's epilogue --
 * the same element update as nadm_adam, bit for bit -- instead of writing the gradient out for a separate launch (which
 * the sample-sharded step still does: its gradients have to be summed over ranks first, nadm_step).  adam == NULL: exactly
 * nadm_decode_bce(_gather) / nadm_encode_bwd.  With adam, m and v point at the Adam moments of the SAME rows as P / V
 * (sub-range launches advance them like P / V), the gradient buffer is left untouched by the matrix-core kernels
 * (K <= 16, C <= 8; the other variants write it and run the update as a second kernel), xg may be NULL. */
typedef struct {
    float*  m;            /* first moment of the rows being updated  */
    float*  v;            /* second moment                            */
    float   lr;
    int32_t step;         /* 1-based step count                       */
    float   grad_scale;   /* gradients are multiplied by this first   */
    int32_t reserved;     /* 0 */
} nadm_adam_t;
int nadm_decode_bce_step(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                         float* P, int32_t kp, const float* Q, int32_t SP,
                         float* dP, float* dqpart, float* losspart, int32_t with_loss,
                         uint8_t* xg, const nadm_adam_t* adam, void* stream);
/* nadm_decode_bce_step (xg and adam may be NULL: then nadm_decode_bce / _gather) with this head's Q operand images from
 * nadm_mlp_fwd_images (kp <= 16): the same bf16 pieces, the same results bit for bit. */
int nadm_decode_bce_images(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                           float* P, int32_t kp, const float* Q, int32_t SP,
                           float* dP, float* dqpart, float* losspart, int32_t with_loss,
                           uint8_t* xg, const nadm_adam_t* adam, const void* qimg, void* stream);
/* The same kernel (qimg may be NULL) with the batch's 64-sample tiles dealt to n_slices blocks per 256-SNP chunk: below ~130k SNPs a
 * launch lasts as long as ONE block's serial chain (prologue + 13 tiles at b = 800: 49 us whether 98 or 196 blocks run), not as long as
 * the chip needs, and pass 2 runs at a third to a half of its rate (profiles/r05_ablations.txt item 11).  dQ rows, the batch copy and everything else per sample are written by the slice that owns the sample; every
 * slice parks its partial of dP (and of the loss value) in `slab`, and the block that is counted last in `counters` adds the
 * partials in slice order and runs the epilogue -- no block waits for another, the result is reproducible bit for bit, and it differs
 * from the n_slices = 1 result by the rounding of that sum only.  n_slices: take nadm_decode_slices(b, M, kp) (1 for kp > 16 and
 * wherever the chunks suffice) -- the plan (nadm_step) and every caller that wants its bits use that function; slab: at least
 * nadm_decode_slab_floats(M, kp, n_slices) floats, 16-byte aligned; counters: one int32 per chunk (nadm_decode_chunks), ZERO-FILLED
 * ONCE by the caller (a launch returns them to zero).  Concurrent launches (heads on two streams) need regions of their own. */
int32_t nadm_decode_slices(int32_t b, int64_t M, int32_t kp);
int32_t nadm_decode_slices_max(int32_t bmax, int64_t M, int32_t kp);   /* the largest value over batches of 1..bmax rows: what sizes a slab */
int64_t nadm_decode_slab_floats(int64_t M, int32_t kp, int32_t n_slices);
int nadm_decode_bce_sliced(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                           float* P, int32_t kp, const float* Q, int32_t SP,
                           float* dP, float* dqpart, float* losspart, int32_t with_loss,
                           uint8_t* xg, const nadm_adam_t* adam, const void* qimg,
                           int32_t n_slices, float* slab, int32_t* counters, void* stream);
#ifdef NADM_TEST_HOOKS   /* the TEST build only (csrc/build.sh -> libnadm_testhooks.so); the shipping library does not export it */
/* n > 0 makes nadm_decode_slices return n (capped by the number of tiles) for every kp <= 16 shape, 0 = the library's choice.
 * Process-wide; set it before the buffers of a plan are sized (a plan refuses a step that would need more slices than its slab holds). */
void nadm_test_force_slices(int32_t n);
#endif
/* `weights` (may be NULL): the MLP weight-gradient partials -- the first half of nadm_mlp_bwd_weights, which like pass 3
 * depends only on the outputs of nadm_mlp_bwd(grad_small = NULL) -- are computed by extra blocks of the same launch (they
 * fill the under-occupied last round of pass 3) into small_part [nadm_sample_splits(b), n_small]; nadm_small_grads then
 * sums them into grad_small and, with Adam state, updates the small parameters in the same launch. */
typedef struct {
    const nadm_heads_t* hd;
    const float *Zn, *H, *dL, *dHpre, *dgp;   /* as for nadm_mlp_bwd_weights */
    float* small_part;
} nadm_mlp_weights_t;
int nadm_encode_bwd_step(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                         const float* dZ, const void* dzimg, int32_t CP, float* V, float* dV, const nadm_adam_t* adam,
                         const nadm_mlp_weights_t* weights, int32_t flags /* NADM_X_CLEAN, see nadm_encode_bwd */, void* stream);
/* Pass 3 with the batch's 128-sample tiles dealt to n_slices blocks per 512-SNP chunk (r06; C <= 8): for launches whose chunks alone leave
 * most CUs idle -- an SNP-sharded rank's 6400 rows x 62.5k SNPs is 122 blocks.  Every slice parks its partial [512 x CP] sum in `slab`
 * (nadm_encode_slab_floats(M, CP, n_slices) floats, 16-byte aligned), the block counted last (`counters`: nadm_encode_bwd_chunks(M) int32,
 * zero-filled once, left zero) adds them in slice order and applies the update / stores the gradient: pass 2's hand-off, reproducible
 * bit for bit.  n_slices comes from nadm_encode_slices(b, M, CP) -- a function of the shape alone, 1 for every single-GPU BASELINE shape --
 * so that every path to the pass cuts the batch the same way; adam / weights as for nadm_encode_bwd_step (both may be NULL). */
int32_t nadm_encode_slices(int32_t b, int64_t M, int32_t CP);
int32_t nadm_encode_slices_max(int32_t bmax, int64_t M, int32_t CP);
int64_t nadm_encode_slab_floats(int64_t M, int32_t CP, int32_t n_slices);
int64_t nadm_encode_bwd_chunks(int64_t M);
int nadm_encode_bwd_sliced(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                           const float* dZ, const void* dzimg, int32_t CP, float* V, float* dV, const nadm_adam_t* adam,
                           const nadm_mlp_weights_t* weights, int32_t flags, int32_t n_slices, float* slab, int32_t* counters, void* stream);
#ifdef NADM_TEST_HOOKS   /* the TEST build only: n > 0 makes nadm_encode_slices return n (capped by the tiles), 0 = the library's choice */
void nadm_test_force_p3_slices(int32_t n);
#endif
int nadm_small_grads(const float* small_part, int32_t splits, int32_t n_small, float* grad_small, float* small,
                     const nadm_adam_t* adam, void* stream);
/* nadm_encode_fwd of the NEXT step with nadm_small_grads of this one riding in the same launch as side blocks (pass 1 reads
 * none of the small parameters; the MLP forward behind it reads the updated ones): same sums, same update, one launch and a
 * launch gap less per step.  The caller owes the update to anything that reads `small` before that next pass 1. CP <= 8. */
int nadm_encode_fwd_small(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                          const float* V, int32_t CP, float* zpart, const float* small_part, int32_t splits, int32_t n_small,
                          float* grad_small, float* small, const nadm_adam_t* adam, void* stream);

/* ---- a11: MLP backward (softmax, Linear, ReLU, RMSNorm) ---------------------------------- */
/* Reduces dqpart (the heads' slabs laid back to back in head order, head h holding
 * nadm_decode_chunks(M,kp_h)*b*kp_h floats), writes dZ [b,CP], the flat small-parameter gradient
 * grad_small [n_small] (NULL: left to nadm_mlp_bwd_weights), and when n_loss>0 adds the step's loss (sum of losspart[0..n_loss)) to
 * loss_acc[0] (running sum) and stores it in loss_acc[1] (last step); loss_acc is double[2].
 * Scratch: dL [b,SP], dHpre [b,Hd], dgp [b,CP], small_part [nadm_sample_splits(b), n_small].  dqpart is IN/OUT scratch of the
 * step (hence not const): a slab of more than 2560 rows is folded to 64 rows IN PLACE first (its first 64 rows then hold
 * partial sums, the others are unchanged); shorter slabs are left untouched.  Do not read dqpart after this call. */
int nadm_mlp_bwd(const nadm_heads_t* hd, const float* small, float* dqpart, int64_t M, int32_t b,
                 const float* Z, const float* rinv, const float* Zn, const float* H, const float* Q,
                 float* dL, float* dHpre, float* dgp, float* small_part,
                 float* dZ, float* grad_small,
                 const float* losspart, int64_t n_loss, double* loss_acc, void* stream);

/* out[e] = sum over r < rows of src[r * n + e]  (fixed order; n a multiple of 4, 16-byte aligned pointers): the fold of a
 * partial slab -- zpart [chunks, b*CP] of nadm_encode_fwd, a head's dqpart [chunks, b*kp] -- to one row.  The SNP-sharded step
 * (8(f)-4) folds its rank-local partial sums with it before the two small all-reduces of a step. */
int nadm_sum_rows(const float* src, int64_t rows, int64_t n, float* out, void* stream);

/* ---- 8(f): supervised mode,  weight * CrossEntropyLoss(sum)(Q_0, labels)  (neural_admixture.py:293,470-473;
 * the reference feeds the softmax OUTPUT of head 0 to CrossEntropyLoss, i.e. a second softmax; default weight 100).
 * labels int32 [rows] = class per resident row (train.py:78-81 mapping), idx = the batch's rows (NULL: labels is
 * already per batch row); n_classes must equal k (train.py:79 asserts it).  Call after head 0's nadm_decode_bce and
 * before nadm_mlp_bwd: ADDS weight*(softmax(q_i) - onehot(y_i)) to chunk 0 of head 0's dQ slab (dqpart0 [.., b, kp])
 * and writes the weighted loss to *loss_slot (one float that nadm_mlp_bwd's n_loss range should cover). */
int nadm_supervised_ce(const float* Q, int32_t SP, int32_t k, int32_t kp, const int32_t* labels, const int32_t* idx,
                       int32_t b, int32_t n_classes, float weight, float* dqpart0, float* loss_slot, void* stream);

/* The weight-gradient half of nadm_mlp_bwd on its own (dWk, dbk, dW1, db1, dg from dL, dHpre, dgp, H, Zn -> grad_small):
 * pass grad_small = NULL to nadm_mlp_bwd and call this on any stream ordered after it -- it is independent of pass 3,
 * so a second stream can run it (and the small Adam) underneath nadm_encode_bwd. */
int nadm_mlp_bwd_weights(const nadm_heads_t* hd, int32_t b, const float* Zn, const float* H, const float* dL,
                         const float* dHpre, const float* dgp, float* small_part, float* grad_small, void* stream);

/* ---- a11: dV = X^T . dZ  (autograd of neural_admixture.py:172) ---------------------------
 * C <= 8 (CP 4, 8) runs on gfx950's FP4 x FP6 block-scaled matrix instruction: the 2-bit codes enter it as FP4 numbers without a
 * conversion, and dZ enters as an OPERAND IMAGE -- per 32 samples x column eight FP6 pieces (the hexadecimal digits of |dZ| in
 * fixed point below the block's maximum) with their E8M0 scales, laid out per lane of the instruction.  The image is the same
 * for every block of the launch, so it is built once: nadm_dz_image(dZ [b, CP] -> dzimg, nadm_dz_image_bytes(b) bytes, 16-byte
 * aligned), on the stream, before the pass.  dZ itself is read by the CP > 8 variants only (dzimg may be NULL for them). */
int64_t nadm_dz_image_bytes(int32_t b);
int64_t nadm_dz_image_tile_bytes(void);      /* bytes of one 128-sample tile of the image */
int nadm_dz_image(const float* dZ, int32_t b, int32_t CP, void* dzimg, void* stream);
/* nadm_mlp_bwd that also leaves dZ's operand image in dzimg: the block that completes a group of 32 samples last builds the group's
 * part (no launch of its own, no launch gap).  dz_counters: (b + 31) / 32 int32 on the device, ZERO-FILLED ONCE by the caller (the
 * launch returns them to zero).  Only the 32-sample groups the batch touches are written: a caller whose batch is shorter than the
 * one before clears the image's last tile first (nadm_dz_image_tile_bytes; nadm_step does), or the groups between the batch and
 * the end of that tile keep the earlier batch's pieces -- harmless while they are finite (pass 3 multiplies them by X = 0), but
 * a group that held an inf carries the NaN scale. */
int nadm_mlp_bwd_image(const nadm_heads_t* hd, const float* small, float* dqpart, int64_t M, int32_t b,
                       const float* Z, const float* rinv, const float* Zn, const float* H, const float* Q,
                       float* dL, float* dHpre, float* dgp, float* small_part, float* dZ, float* grad_small,
                       const float* losspart, int64_t n_loss, double* loss_acc, void* dzimg, int32_t* dz_counters, void* stream);
#define NADM_X_CLEAN 1     /* flags: xp is nadm_decode_bce_gather's copy of the batch (tiled, rows in batch order, missing calls already 0) */
int64_t nadm_batch_copy_bytes(int32_t b, int64_t M);
int nadm_encode_bwd(const uint8_t* xp, int64_t ld, const int32_t* idx, int32_t b, int64_t M,
                    const float* dZ, const void* dzimg, int32_t CP, float* dV, int32_t flags, void* stream);

/* ---- a12/a13: Adam(betas .9/.95, eps 1e-8) + restrict_P (neural_admixture.py:187-204,411-412)
 * Flat update of n floats; `step` is the 1-based step count; gradients are multiplied by
 * grad_scale first (1/world for the DDP mean, neural_admixture.py:317); elements with index
 * >= clamp_from are clamped to [0,1] after the update (pass n for "no clamp"). */
int nadm_adam(float* param, const float* grad, float* m, float* v, int64_t n, int64_t clamp_from,
              float lr, int32_t step, float grad_scale, void* stream);

/* ======================================================================================================================
 * The training step as ONE call  (replaces NeuralAdmixture._run_step + the DDP hooks + optimizer.step + restrict_P,
 * neural_admixture.py:315-319,403-414, as one host call per step; the entry points above remain the pieces it is made of)
 * ======================================================================================================================
 * Flat parameter layout.  Parameters, gradients and (single / SNP-sharded mode) Adam moments live in ONE flat float buffer each:
 *     [ small | pad | V [M,CP] | gap | P_0 [M,KP_0] | P_1 ... | gap ]
 * cut into the two MESSAGES of the sample-sharded step: B = [0, msg_a_off) holds the small parameters and V, A =
 * [msg_a_off, msg_a_off + world * slice_a) holds every head's P; the gaps (zeros, < 4 * world floats) make both a multiple of
 * `world` slices.  world = 1: no gaps, pad = to a multiple of 64 floats.
 * Message B travels as n_buckets BUCKETS (r05; DDP's bucketed exchange, neural_admixture.py:315-319): bucket j = the floats
 * [bkt_off[j], bkt_off[j+1]) = the V rows of the SNP range [bkt_m0[j], bkt_m0[j+1]) -- bucket 0 also the small parameters in front of
 * them -- as `world` contiguous slices of bkt_slice[j] floats (range-major: every bucket is a message of its own, reduce-scattered,
 * updated and all-gathered while the other ranges are still being computed or already consumed).  Range boundaries are multiples of
 * every kernel's chunk (2048 SNPs) AND of what makes a bucket 4 * world floats long; a request for more buckets than M allows yields
 * fewer (n_buckets is the number actually cut; 1 = the single message of r04).  slice_b = sum of bkt_slice = floats of B per rank;
 * the rank's moments of bucket j start at bkt_mom[j] in its moment buffers (DP mode: [slice_b | slice_a] floats each). */
typedef struct nadm_flat_layout {
    int64_t n_flat;                       /* floats in a flat buffer                                  */
    int64_t off_v;                        /* V starts here (n_small rounded up to lcm(64, 4 * world)) */
    int64_t off_p[NADM_MAX_HEADS];        /* head h's P starts here                                   */
    int64_t slice_b, slice_a;             /* floats per rank of message B (all buckets) / A           */
    int64_t msg_a_off;                    /* where message A starts = end of the last bucket of B     */
    int32_t n_buckets, reserved;          /* buckets message B is cut into (>= 1)                     */
    int64_t bkt_off[NADM_MAX_BUCKETS + 1];/* bucket j = floats [bkt_off[j], bkt_off[j+1]); [n] = msg_a_off */
    int64_t bkt_slice[NADM_MAX_BUCKETS];  /* floats per rank of bucket j                              */
    int64_t bkt_m0[NADM_MAX_BUCKETS + 1]; /* bucket j holds the V rows of SNPs [bkt_m0[j], bkt_m0[j+1]); [n] = M */
    int64_t bkt_mom[NADM_MAX_BUCKETS];    /* start of the rank's slice of bucket j in its B moments   */
} nadm_flat_layout_t;
int nadm_flat_layout(const nadm_heads_t* hd, int64_t M, int32_t world, int32_t n_buckets, nadm_flat_layout_t* out);

/* Transport of the step's collectives.  The step issues them on hipStreams it names; an implementation must order its work on
 * that stream (RCCL does; a host transport synchronises the stream).  All three operate IN PLACE on float buffers:
 *   reduce_scatter(buf, slice): buf holds world * slice floats; afterwards THIS rank's slice [rank*slice, (rank+1)*slice) holds the
 *                               sum over ranks of that slice (the other slices are scratch)                    (ncclReduceScatter)
 *   all_gather(buf, slice):     every rank contributes its own slice; afterwards buf is complete everywhere    (ncclAllGather)
 *   all_reduce(buf, n):         sum over ranks                                                                 (ncclAllReduce)
 *   async_error():              (may be NULL) nonzero + nadm_last_error() if a collective queued earlier has failed asynchronously
 *                               (ncclCommGetAsyncError); nadm_plan_flush asks, and every step of a plan built with debug != 0
 * nadm_comm_rccl: a communicator of its own over RCCL / xGMI (ncclCommInitRank with `unique_id`, 128 bytes from
 * nadm_comm_rccl_unique_id on rank 0, distributed by the caller -- the reference gets its communicator from
 * init_process_group("nccl"), src/utils.py:88-93).  librccl_path: the librccl.so to dlopen (NULL: "librccl.so"); pass the path of
 * the copy already mapped into the process.  ncclCommInitRank is itself a collective -- it returns when EVERY rank has entered it --
 * so it runs under a watchdog: after timeout_ms (<= 0: wait forever) the call gives up with status 5 and a message naming the
 * rank; the helper thread stays blocked inside the library, i.e. the process is expected to tear down (the reference's behaviour on
 * any rank failure: teardown + re-raise, src/main.py:119-133).  nadm_comm_rccl_probe: dlopen + symbol resolution only -- what can
 * fail without any other rank being involved; callers agree on its outcome BEFORE anyone enters ncclCommInitRank (comm.py).
 * nadm_comm_emulated: rank 0 of `world` ranks with no-op collectives -- the per-rank cost of a world-rank step measured on ONE GPU
 * (bench.py --emulate-world; the run's results are meaningless). */
typedef struct nadm_comm {
    int32_t rank, world;
    void* ctx;
    int (*reduce_scatter)(void* ctx, float* buf, int64_t slice, void* stream);
    int (*all_gather)(void* ctx, float* buf, int64_t slice, void* stream);
    int (*all_reduce)(void* ctx, float* buf, int64_t n, void* stream);
    void (*destroy)(void* ctx);           /* may be NULL */
    int (*async_error)(void* ctx);        /* may be NULL */
} nadm_comm_t;
int  nadm_comm_rccl_probe(const char* librccl_path);
int  nadm_comm_rccl_unique_id(const char* librccl_path, void* id128);
int  nadm_comm_rccl(const char* librccl_path, const void* id128, int32_t rank, int32_t world, int32_t timeout_ms, nadm_comm_t** out);
int  nadm_comm_emulated(int32_t world, nadm_comm_t** out);
void nadm_comm_free(nadm_comm_t* comm);
/* ncclCommAbort instead of ncclCommDestroy: frees a communicator whose collectives may never complete (a peer is gone) */
void nadm_comm_abort(nadm_comm_t* comm);

/* How the work of a step is spread over ranks */
#define NADM_MODE_SINGLE 0   /* one GPU: Adam + restrict_P in the epilogues of passes 2 and 3                                        */
#define NADM_MODE_DP     1   /* samples sharded (the reference's scheme, neural_admixture.py:287,315-319): gradients summed over ranks,
                              * optimizer SHARDED -- reduce-scatter -> Adam (grad_scale 1/world = DDP's mean) + restrict_P on this rank's
                              * 1/world slice of the parameters and ITS moments only -> all-gather of the updated parameters.  Same wire
                              * bytes as the all-reduce, optimizer traffic and Adam-moment memory / world.  Message A (all P) travels on a
                              * side stream underneath the MLP backward, pass 3 and the next step's pass 1; message B (small | V) on the
                              * compute stream behind pass 3 (n_buckets <= 1, the default: the next pass 1 needs all of V, nothing is left
                              * to hide it under) or, n_buckets > 1, as SNP ranges on a second side stream: pass 3 is launched range by
                              * range, behind each range its bucket is reduce-scattered, updated and all-gathered, and the NEXT step's
                              * pass 1 runs as the same ranges, each waiting for its own bucket only.  What that buys is wire time hidden
                              * under pass 3 and pass 1 (~40 us each at 800 rows); what it costs a rank is measured
                              * (profiles/r05_rank_emulation.txt: every cross-stream hand-off is ~10 us on the GPU's timeline).  Issue order
                              * on the communicator (identical on every rank): reduce-scatter A, then per bucket reduce-scatter B_j,
                              * all-gather B_j, then all-gather A -- B, which the next pass 1 waits for, never queues behind A's
                              * all-gather.  comm_a != NULL gives message A a communicator of its own: A and B then share the links
                              * instead of queueing behind each other */
#define NADM_MODE_SNP    2   /* SNPs sharded (8(f)-4): every rank owns M/world SNPs of X, V, P and their Adam state and processes the
                              * GLOBAL batch; two small all-reduces per step (partial Z, partial dQ), Adam in the epilogues with 1/world */

/* Everything the step touches; every pointer is caller-owned device memory (sizes: the entry points above) */
typedef struct nadm_plan_desc {
    int32_t mode, bmax;
    int64_t M, ld;
    nadm_heads_t heads;
    const uint8_t* xp;                    /* resident packed rows (nadm_plan_set_rows replaces it)                             */
    float *params, *grads;                /* flat buffers, nadm_flat_layout(heads, M, comm world in DP mode, else 1)            */
    float *m, *v;                         /* Adam moments: flat like params (single, SNP); DP: [slice_b | slice_a] of THIS rank */
    float *zpart, *Z, *rinv, *Zn, *H, *Q, *dL, *dHpre, *dgp, *dZ, *dqpart, *losspart, *small_part;
    float *zsum, *dqsum;                  /* SNP mode: [bmax*CP], [bmax*SP]                                                     */
    void* qimg; int64_t qimg_head_bytes;  /* nadm_mlp_fwd_images (zero-filled once)                                            */
    void* dzimg; int32_t* dzcnt;          /* nadm_mlp_bwd_image  (C <= 8; counters zero-filled once)                            */
    uint8_t* xg;                          /* nadm_batch_copy_bytes(bmax, M) (C <= 8)                                            */
    double* loss_acc;                     /* [2]: running sum, last step                                                        */
    const nadm_comm_t* comm;              /* NULL: one rank.  Must outlive the plan                                             */
    const nadm_comm_t* comm_a;            /* DP mode: a second communicator for message A (NULL: `comm` carries both)           */
    int32_t n_buckets;                    /* DP mode: buckets of message B (0, 1: one message; params / grads / m / v are laid
                                           * out by nadm_flat_layout(heads, M, world, n_buckets))                               */
    int32_t p3_whole;                     /* DP mode, n_buckets > 1: != 0 keeps pass 3 ONE launch (the buckets' collectives then
                                           * all start behind it; only the next pass 1 is pipelined against them)                */
    int32_t debug;                        /* != 0: every step asks the communicators for asynchronous errors (a host call each)  */
    int32_t reserved;                     /* 0 */
    float* p2_slab;                       /* pass 2 in sample slices (nadm_decode_bce_sliced): head h's region starts at the sum over the
                                           * heads before it of nadm_decode_slab_floats(M, kp, nadm_decode_slices_max(bmax, M, kp)); NULL (with
                                           * p2_cnt): pass 2 is never sliced                                                        */
    int32_t* p2_cnt;                      /* the heads' counters back to back, nadm_decode_chunks(M, kp) each, zero-filled once    */
    float* p3_slab;                       /* pass 3 in sample slices (nadm_encode_bwd_sliced): nadm_encode_slab_floats(M, CP,
                                           * nadm_encode_slices_max(bmax, M, CP)) floats; NULL (with p3_cnt): pass 3 is never sliced   */
    int32_t* p3_cnt;                      /* nadm_encode_bwd_chunks(M) counters, zero-filled once                                    */
} nadm_plan_desc_t;
typedef struct nadm_plan nadm_plan_t;
int  nadm_plan_create(const nadm_plan_desc_t* desc, nadm_plan_t** out);
void nadm_plan_destroy(nadm_plan_t* plan);
int  nadm_plan_set_rows(nadm_plan_t* plan, const uint8_t* xp);
/* supervised mode (neural_admixture.py:460-474): class per resident row, NULL switches it off */
int  nadm_plan_set_labels(nadm_plan_t* plan, const int32_t* labels, int32_t n_classes, float weight);
/* Adam step count (1-based count of completed steps; 0 after loading parameters) and whether every P entry lies in [0, 1] (true
 * after any step -- restrict_P -- and for the GMM initialisation; while false the loss path clamps the reconstruction before the
 * logarithm, nadm_decode_bce with_loss bit 1).  Callers that run optimizer steps of their own (nadm_adam) keep the count in step. */
int  nadm_plan_set_state(nadm_plan_t* plan, int32_t step_count, int32_t p_in_unit_range);
int32_t nadm_plan_p_in_unit_range(const nadm_plan_t* plan);
int32_t nadm_plan_step_count(const nadm_plan_t* plan);

/* ONE training step on the batch rows idx[0..b) -- gather/decode, forward, loss, backward, gradient exchange, Adam, restrict_P --
 * queued on `stream` (and, in DP mode, on the plan's side streams), asynchronous.  with_loss: also add the step's loss value to
 * loss_acc.  The step may leave work to the NEXT step's launches (single / SNP: the small-parameter update rides in the next
 * pass 1; DP: the messages complete underneath the next step's first launches): nadm_plan_flush makes `stream` see every
 * parameter final (and reports a collective that failed asynchronously).
 * A step that FAILS part-way (a refused launch, a transport error) leaves the plan POISONED: the Adam step count, the pending
 * hand-offs and the side streams are in no defined state, and every later nadm_step / nadm_plan_flush / nadm_plan_infer on it
 * fails fast with a message that says so.  Destroy the plan and build a new one (parameters and moments live in the caller's
 * buffers; what the failed step did to them is undefined). */
int  nadm_step(nadm_plan_t* plan, const int32_t* idx, int32_t b, float lr, int32_t with_loss, void* stream);
int  nadm_plan_flush(nadm_plan_t* plan, void* stream);
/* Encoder only (final Q, neural_admixture.py:369-383; inference.py:71-77): pass 1 + MLP forward -> Q [b, SP] */
int  nadm_plan_infer(nadm_plan_t* plan, const int32_t* idx, int32_t b, void* stream);

/* Measurement: HIP events around the launches of a step on the stream they run on.  mask bit i = NADM_T_* below; the records cost
 * ~1 % of a step per bit, hence a mask.  nadm_plan_kernel_ms synchronises, writes the mean duration [ms] of every timed group since
 * the last call (0 where nothing was timed) and clears the records. */
#define NADM_T_ENCODE_FWD 0
#define NADM_T_MLP_FWD    1
#define NADM_T_DECODE_BCE 2
#define NADM_T_MLP_BWD    3
#define NADM_T_ENCODE_BWD 4
#define NADM_T_SYNC_A     5   /* DP: message A on the side stream: from its reduce-scatter to its all-gather, which is issued behind message B (a span) */
#define NADM_T_SYNC_B     6   /* DP: message B on its side stream: from the first bucket's small-gradient sum + reduce-scatter to the last bucket's all-gather (a span) */
#define NADM_T_COUNT      7
int  nadm_plan_timing(nadm_plan_t* plan, uint32_t mask);
int  nadm_plan_kernel_ms(nadm_plan_t* plan, float* ms /* [NADM_T_COUNT] */, int32_t* counts /* [NADM_T_COUNT], may be NULL */);
/* With NADM_T_SYNC_B in the mask: the mean duration [ms] of every BUCKET of message B on its side stream (reduce-scatter -> Adam ->
 * all-gather of the range), bucket j in ms[j]; returns through *n the number of buckets.  Call before nadm_plan_kernel_ms (which
 * clears the records). */
int  nadm_plan_bucket_ms(nadm_plan_t* plan, float* ms /* [NADM_MAX_BUCKETS] */, int32_t* n);
int32_t nadm_plan_poisoned(const nadm_plan_t* plan);

#ifdef NADM_TEST_HOOKS   /* the TEST build only: route nadm_mlp_fwd / nadm_mlp_bwd to the generic (any hidden width) kernels even where the
                          * register-resident ones apply, so that tests can compare the two.  Process-wide. */
void nadm_test_force_generic_mlp(int32_t on);
#endif

/* ---- 8(f)-3: log-likelihood report from the packed matrix (src/utils_c/utils.pyx:15-40, called train.py:134-146) -----
 * partial[b] (b < nadm_loglik_blocks(M) = 8 row slices x ceil(M / 1024), double, device) = sum over the block's 1024 SNPs and
 * its slice of the `rows` rows of g*log(rec) + (2-g)*log1p(-rec) over non-missing calls, rec = clip(Q_i.P_j, eps, 1-eps), g = clip(code, eps, 2-eps), all
 * in float64 like the reference.  P [M,K] float32 (unpadded, the returned Ps[i]), Q [rows, >=K] float32 with row stride
 * q_stride, both on the device; K <= 16; eps in [1e-9, 0.5) (the reference's: 1e-6).  The caller adds the partials (fixed order).
 * (The logarithms are taken of products over 16 genotypes: the same float64 sum to ~1e-15 relative.) */
int64_t nadm_loglik_blocks(int64_t M);
int nadm_loglik(const uint8_t* xp, int64_t ld, int64_t rows, int64_t M, const float* P, const float* Q, int32_t K,
                int32_t q_stride, double eps, double* partial, void* stream);

/* ---- 8(f)-3: decoder init, the means of the mixture the reference fits in the PCA subspace (model/train.py:61-66, scikit-learn's
 * GaussianMixture(n_components=K, n_init=5, init_params='k-means++', tol=1e-4, covariance_type='full', max_iter=100,
 * random_state=seed).fit(X).means_): the EM iterations of that call in float64 on the host (csrc/nadm_gmm.cpp restates the
 * algorithm and the library's conventions).  X [N, d] row-major; picks [n_init, K] = the k-means++ seed rows of every restart,
 * drawn by the caller from the library's own random stream (gmm.kmeanspp_picks); means [K, d] out; lower_bound / n_iter (may be
 * NULL) = the winning restart's objective and iteration count.  Host code, one thread per restart, synchronous. */
int nadm_gmm_fit_means(const double* X, int64_t N, int32_t d, int32_t K, const int32_t* picks, int32_t n_init, double tol,
                       int32_t max_iter, double reg_covar, double* means, double* lower_bound, int32_t* n_iter);
/* The same fit with the sums over the samples on the device (csrc/nadm_gmm_dev.hip: the restarts side by side in one grid, five small
 * launches per EM iteration, float64, fixed summation order): for inputs where the host form is a visible part of a run (0.77 s at
 * N = 100k, here ~0.03 s).  d == 8 (the reference's --pca_components default) and K <= 16 only.  X, picks, means are HOST pointers
 * like above (X is uploaded: 64 B per sample); synchronous; work is queued on `stream`.  Equal to the host form up to the order of
 * the sums (means to ~1e-12). */
int nadm_gmm_fit_means_dev(const double* X, int64_t N, int32_t d, int32_t K, const int32_t* picks, int32_t n_init, double tol,
                           int32_t max_iter, double reg_covar, double* means, double* lower_bound, int32_t* n_iter, void* stream);

/* ---- VCF genotypes (src/snp_reader.py:73-87: scikit-allel read_vcf, calldata/GT as int8 with -1 fills, summed over the two
 * alleles, negative sums -> 3).  buf = the whole decompressed file; out == NULL only counts samples and variant lines;
 * out = uint8 [n_samples, n_variants], sample-major like the reference's matrix.  Host code (threads), no GPU. */
int nadm_vcf_parse_gt(const char* buf, int64_t len, int64_t* n_samples, int64_t* n_variants, uint8_t* out);

/* ---- 8(f)-4: ADMIXTURE-compatible text output ------------------------------------------------------------
 * np.savetxt(path, A, delimiter=' ') for a float32 host matrix, byte for byte ('%.18e' of the value widened to
 * double, src/utils.py:56-66); multi-threaded.  a [rows, cols] with row stride row_stride (elements). */
int nadm_savetxt_f32(const char* path, const float* a, int64_t rows, int64_t cols, int64_t row_stride);

/* ---- measurement helpers ------------------------------------------------------------------ */
/* Synthetic admixture-model genotypes written directly as packed bytes:
 * G ~ Binomial(2, Qt.F), `missing` fraction set to 3.  Qt [rows,K], Fq [K,M] float32 on device;
 * counter-based RNG keyed by (seed, row0 + r, SNP) so shards generated on different ranks agree. */
int nadm_synth_packed(uint8_t* xp, int64_t rows, int64_t row0, int64_t M, int64_t ld,
                      const float* Qt, const float* Fq, int32_t K, float missing, uint64_t seed, void* stream);

/* Box fingerprint (bench.py "box"; csrc/nadm_calib.hip): a fixed stream of packed-f32 VALU chains + bf16 matrix instructions, three
 * waves per SIMD on every CU, `iters` rounds.  Block i writes out[2i] = shader cycles (s_memtime) and out[2i+1] = constant-rate ticks
 * (s_memrealtime, nadm_wall_clock_khz) it took: cycles / ticks x rate = the shader clock the box sustains under an issue-bound load.
 * Returns the number of reporting blocks (<= max_blocks), or a negative status.  `sink`: one float, never written. */
/* Measurement only: with a non-NULL device pointer every following S = 1 launch of the matrix-core pass 2 has the block in the middle of its
 * grid write out2[0] = shader cycles (s_memtime) and out2[1] = constant-rate ticks (s_memrealtime) it took: the clock the DOMINANT kernel
 * itself ran at in this run.  NULL switches it off.  Process-wide (one process per GPU); a few scalar instructions in one block. */
void nadm_clock_probe(uint64_t* out2_dev);
int nadm_calib_clock(int32_t iters, uint64_t* out /* device [2 * max_blocks] */, int32_t max_blocks, float* sink, void* stream);
int64_t nadm_wall_clock_khz(void);

#ifdef __cplusplus
}
#endif
#endif /* NADM_H */
