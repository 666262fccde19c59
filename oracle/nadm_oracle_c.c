/*
 * nadm_oracle_c.c -- plain C (OpenMP) restatement of ONE Neural ADMIXTURE training step on the CPU.
 *
 * TEST INFRASTRUCTURE ONLY: this is the timed "cpu_baseline" (kind "port") of bench.py and a second
 * checker for tests/; the product never links or calls it.  It is pinned against the same golden
 * vectors as oracle/nadm_oracle.py (tests/test_oracle_c.py).
 *
 * Algorithm restated (reference paths relative to /root/reference/neural_admixture):
 *   X = G/2, missing(3) -> 0                          model/neural_admixture.py:169-170
 *   Z = X V ; RMSNorm(eps 1e-8, weight g)             :172-173, :135
 *   H = relu(Zn W1^T + b1) ; Q = softmax(H Wk^T + bk) :174-176 (single head here: K given)
 *   R = clamp(Q P^T, 0, 1) ; loss = BCE(sum)(R, X)    :94-97, :288, :431
 *   backward (closed form of autograd), Adam(.9,.95,1e-8), clamp P to [0,1]   :410-412, :187-204
 * Unlike the reference's ATen graph it never materialises a [b,M] fp32 tensor (the reference spends
 * ~90 % of its CPU time on exactly those temporaries), so it is a STRONGER
 * CPU baseline than the reference's own path.
 *
 * Layouts: G uint8 [N,M] row-major (unpacked, as the reference's CPU path keeps it), batch rows by
 * index; V [M,C]; P [M,K] (SNP-major); small params as separate arrays.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define MAXC 32
#define MAXK 64

typedef struct {
    int64_t M;
    int C, K, Hd;
    float *V, *P;                 /* [M,C], [M,K] */
    float *g, *W1, *b1, *Wk, *bk; /* [C], [Hd,C], [Hd], [K,Hd], [K] */
    /* Adam moments, same shapes */
    float *mV, *vV, *mP, *vP, *mg, *vg, *mW1, *vW1, *mb1, *vb1, *mWk, *vWk, *mbk, *vbk;
    int step;
} oracle_model_t;

static inline float decode_x(uint8_t g) { return g == 3 ? 0.0f : 0.5f * (float)g; }

static void adam_update(float* p, const float* g, float* m, float* v, int64_t n, float lr, int step, int clamp01) {
    const double bc1 = 1.0 - pow(0.9, step), bc2 = 1.0 - pow(0.95, step);
    const float step_size = (float)(lr / bc1), bc2s = (float)sqrt(bc2);
    const float omb1 = (float)(1.0 - 0.9), omb2 = (float)(1.0 - 0.95);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float gr = g[i];
        m[i] = m[i] + (gr - m[i]) * omb1;
        v[i] = v[i] * 0.95f + gr * gr * omb2;
        const float den = sqrtf(v[i]) / bc2s + 1e-8f;
        float q = p[i] - step_size * (m[i] / den);
        if (clamp01) q = q < 0.f ? 0.f : (q > 1.f ? 1.f : q);
        p[i] = q;
    }
}

/* One training step on rows idx[0..b) of G.  Returns the BCE loss (double).
 * If grads_out != NULL it receives dV[M*C] | dP[M*K] | dg[C] | dW1[Hd*C] | db1[Hd] | dWk[K*Hd] | dbk[K]
 * (for parity tests); if apply != 0 the Adam update + P clamp are applied. Also exports Q [b,K] if Q_out. */
double oracle_train_step(oracle_model_t* mdl, const uint8_t* G, const int64_t* idx, int b, float lr, int apply,
                         float* grads_out, float* Q_out) {
    const int64_t M = mdl->M;
    const int C = mdl->C, K = mdl->K, Hd = mdl->Hd;
    const int nt = omp_get_max_threads();
    float* Z = (float*)calloc((size_t)b * C, sizeof(float));
    float* Zn = (float*)malloc((size_t)b * C * sizeof(float));
    float* rinv = (float*)malloc((size_t)b * sizeof(float));
    float* H = (float*)malloc((size_t)b * Hd * sizeof(float));
    float* Q = (float*)malloc((size_t)b * K * sizeof(float));
    float* dQ = (float*)calloc((size_t)b * K, sizeof(float));
    float* dZ = (float*)malloc((size_t)b * C * sizeof(float));
    float* dV = (float*)malloc((size_t)M * C * sizeof(float));
    float* dP = (float*)malloc((size_t)M * K * sizeof(float));
    float* tpriv = (float*)calloc((size_t)nt * b * (C > K ? C : K), sizeof(float));
    const int W = C > K ? C : K;

    /* ---- pass 1: Z = X V, thread-private partials over SNP blocks ---- */
#pragma omp parallel
    {
        float* zp = tpriv + (size_t)omp_get_thread_num() * b * W;
        memset(zp, 0, (size_t)b * W * sizeof(float));
#pragma omp for schedule(static)
        for (int64_t m0 = 0; m0 < M; m0 += 256) {
            const int64_t m1 = m0 + 256 < M ? m0 + 256 : M;
            for (int i = 0; i < b; ++i) {
                const uint8_t* row = G + idx[i] * M;
                float acc[MAXC];
                for (int c = 0; c < C; ++c) acc[c] = 0.f;
                for (int64_t m = m0; m < m1; ++m) {
                    const float x = decode_x(row[m]);
                    const float* v = mdl->V + m * C;
                    for (int c = 0; c < C; ++c) acc[c] += x * v[c];
                }
                for (int c = 0; c < C; ++c) zp[i * W + c] += acc[c];
            }
        }
    }
    for (int t = 0; t < nt; ++t)
        for (int i = 0; i < b; ++i)
            for (int c = 0; c < C; ++c) Z[i * C + c] += tpriv[((size_t)t * b + i) * W + c];

    /* ---- MLP forward ---- */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < b; ++i) {
        float ms = 0.f;
        for (int c = 0; c < C; ++c) ms += Z[i * C + c] * Z[i * C + c];
        const float ri = 1.0f / sqrtf(ms / (float)C + 1e-8f);
        rinv[i] = ri;
        for (int c = 0; c < C; ++c) Zn[i * C + c] = Z[i * C + c] * ri * mdl->g[c];
        float logit[MAXK];
        for (int k = 0; k < K; ++k) logit[k] = mdl->bk[k];
        for (int h = 0; h < Hd; ++h) {
            float a = mdl->b1[h];
            for (int c = 0; c < C; ++c) a += Zn[i * C + c] * mdl->W1[h * C + c];
            a = a > 0.f ? a : 0.f;
            H[(size_t)i * Hd + h] = a;
            for (int k = 0; k < K; ++k) logit[k] += a * mdl->Wk[k * Hd + h];
        }
        float mxv = logit[0];
        for (int k = 1; k < K; ++k) mxv = logit[k] > mxv ? logit[k] : mxv;
        float s = 0.f;
        for (int k = 0; k < K; ++k) { logit[k] = expf(logit[k] - mxv); s += logit[k]; }
        for (int k = 0; k < K; ++k) Q[i * K + k] = logit[k] / s;
    }
    if (Q_out) memcpy(Q_out, Q, (size_t)b * K * sizeof(float));

    /* ---- pass 2: decoder + BCE fwd/bwd; dP per SNP, dQ thread-private ---- */
    double loss = 0.0;
#pragma omp parallel reduction(+ : loss)
    {
        float* dq = tpriv + (size_t)omp_get_thread_num() * b * W;
        memset(dq, 0, (size_t)b * W * sizeof(float));
#pragma omp for schedule(static)
        for (int64_t m0 = 0; m0 < M; m0 += 64) {
            const int64_t m1 = m0 + 64 < M ? m0 + 64 : M;
            float dp[64 * MAXK];
            memset(dp, 0, sizeof(float) * 64 * K);
            for (int i = 0; i < b; ++i) {
                const uint8_t* row = G + idx[i] * M;
                const float* q = Q + i * K;
                float* dqi = dq + i * W;
                for (int64_t m = m0; m < m1; ++m) {
                    const float* p = mdl->P + m * K;
                    const float x = decode_x(row[m]);
                    float rr = 0.f;
                    for (int k = 0; k < K; ++k) rr += q[k] * p[k];
                    const float r = rr < 0.f ? 0.f : (rr > 1.f ? 1.f : rr);
                    float l1 = logf(r), l0 = log1pf(-r);
                    l1 = l1 < -100.f ? -100.f : l1;
                    l0 = l0 < -100.f ? -100.f : l0;
                    loss -= (double)(x * l1 + (1.f - x) * l0);
                    float den = (1.f - r) * r;
                    den = den < 1e-12f ? 1e-12f : den;
                    float dr = (r - x) / den;
                    if (rr < 0.f || rr > 1.f) dr = 0.f;
                    float* dpm = dp + (m - m0) * K;
                    for (int k = 0; k < K; ++k) { dpm[k] += dr * q[k]; dqi[k] += dr * p[k]; }
                }
            }
            memcpy(dP + m0 * K, dp, sizeof(float) * (size_t)(m1 - m0) * K);
        }
    }
    for (int t = 0; t < nt; ++t)
        for (int i = 0; i < b; ++i)
            for (int k = 0; k < K; ++k) dQ[i * K + k] += tpriv[((size_t)t * b + i) * W + k];

    /* ---- MLP backward ---- */
    float* dWk = (float*)calloc((size_t)K * Hd, sizeof(float));
    float* dW1 = (float*)calloc((size_t)Hd * C, sizeof(float));
    float* db1 = (float*)calloc((size_t)Hd, sizeof(float));
    float dbk[MAXK] = {0}, dg[MAXC] = {0};
    float* dHp = (float*)malloc((size_t)b * Hd * sizeof(float));
    float* dL = (float*)malloc((size_t)b * K * sizeof(float));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < b; ++i) {
        float dot = 0.f;
        for (int k = 0; k < K; ++k) dot += dQ[i * K + k] * Q[i * K + k];
        for (int k = 0; k < K; ++k) dL[i * K + k] = Q[i * K + k] * (dQ[i * K + k] - dot);
        float dzn[MAXC];
        for (int c = 0; c < C; ++c) dzn[c] = 0.f;
        for (int h = 0; h < Hd; ++h) {
            float a = 0.f;
            for (int k = 0; k < K; ++k) a += dL[i * K + k] * mdl->Wk[k * Hd + h];
            a = H[(size_t)i * Hd + h] > 0.f ? a : 0.f;
            dHp[(size_t)i * Hd + h] = a;
            for (int c = 0; c < C; ++c) dzn[c] += a * mdl->W1[h * C + c];
        }
        const float ri = rinv[i];
        float tz = 0.f;
        for (int c = 0; c < C; ++c) tz += dzn[c] * mdl->g[c] * Z[i * C + c];
        tz /= (float)C;
        for (int c = 0; c < C; ++c) {
            dZ[i * C + c] = ri * dzn[c] * mdl->g[c] - Z[i * C + c] * ri * ri * ri * tz;
        }
    }
    /* weight grads (parallel over hidden units) */
#pragma omp parallel for schedule(static)
    for (int h = 0; h < Hd; ++h) {
        for (int i = 0; i < b; ++i) {
            const float d = dHp[(size_t)i * Hd + h], hv = H[(size_t)i * Hd + h];
            db1[h] += d;
            for (int c = 0; c < C; ++c) dW1[h * C + c] += d * Zn[i * C + c];
            for (int k = 0; k < K; ++k) dWk[k * Hd + h] += dL[i * K + k] * hv;
        }
    }
    for (int i = 0; i < b; ++i) {
        for (int k = 0; k < K; ++k) dbk[k] += dL[i * K + k];
        /* dZn[c] = sum_h dHp[i][h] W1[h][c] recomputed for dg (tiny) */
        for (int c = 0; c < C; ++c) {
            float dzn = 0.f;
            for (int h = 0; h < Hd; ++h) dzn += dHp[(size_t)i * Hd + h] * mdl->W1[h * C + c];
            dg[c] += dzn * Z[i * C + c] * rinv[i];
        }
    }

    /* ---- pass 3: dV = X^T dZ ---- */
#pragma omp parallel for schedule(static)
    for (int64_t m0 = 0; m0 < M; m0 += 64) {
        const int64_t m1 = m0 + 64 < M ? m0 + 64 : M;
        float acc[64 * MAXC];
        memset(acc, 0, sizeof(float) * 64 * C);
        for (int i = 0; i < b; ++i) {
            const uint8_t* row = G + idx[i] * M;
            const float* dz = dZ + i * C;
            for (int64_t m = m0; m < m1; ++m) {
                const float x = decode_x(row[m]);
                float* a = acc + (m - m0) * C;
                for (int c = 0; c < C; ++c) a[c] += x * dz[c];
            }
        }
        memcpy(dV + m0 * C, acc, sizeof(float) * (size_t)(m1 - m0) * C);
    }

    if (grads_out) {
        float* o = grads_out;
        memcpy(o, dV, sizeof(float) * M * C); o += M * C;
        memcpy(o, dP, sizeof(float) * M * K); o += M * K;
        memcpy(o, dg, sizeof(float) * C); o += C;
        memcpy(o, dW1, sizeof(float) * Hd * C); o += (size_t)Hd * C;
        memcpy(o, db1, sizeof(float) * Hd); o += Hd;
        memcpy(o, dWk, sizeof(float) * K * Hd); o += (size_t)K * Hd;
        memcpy(o, dbk, sizeof(float) * K);
    }
    if (apply) {
        mdl->step += 1;
        adam_update(mdl->V, dV, mdl->mV, mdl->vV, M * C, lr, mdl->step, 0);
        adam_update(mdl->P, dP, mdl->mP, mdl->vP, M * K, lr, mdl->step, 1);
        adam_update(mdl->g, dg, mdl->mg, mdl->vg, C, lr, mdl->step, 0);
        adam_update(mdl->W1, dW1, mdl->mW1, mdl->vW1, (int64_t)Hd * C, lr, mdl->step, 0);
        adam_update(mdl->b1, db1, mdl->mb1, mdl->vb1, Hd, lr, mdl->step, 0);
        adam_update(mdl->Wk, dWk, mdl->mWk, mdl->vWk, (int64_t)K * Hd, lr, mdl->step, 0);
        adam_update(mdl->bk, dbk, mdl->mbk, mdl->vbk, K, lr, mdl->step, 0);
    }
    free(Z); free(Zn); free(rinv); free(H); free(Q); free(dQ); free(dZ); free(dV); free(dP); free(tpriv);
    free(dWk); free(dW1); free(db1); free(dHp); free(dL);
    return loss;
}

int oracle_num_threads(void) { return omp_get_max_threads(); }
void oracle_set_threads(int n) { omp_set_num_threads(n); }
