#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ v4f mm(v8i a, v8i b, int sa, int sb) {
    v4f c = {0.f, 0.f, 0.f, 0.f};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 0, 0, sa, 0, sb);
}
// out layout: [0..63]: test1 D[lane][0]; [64..127]: test scale; then A-impulse rows [2048], B-impulse cols [2048], match table [2048*16]
__global__ void k(float* f_out, int* arow, int* bcol, int* match, int* nmatch) {
    const int l = threadIdx.x;
    v8i ones_a, ones_b, z;
    for (int r = 0; r < 8; ++r) { ones_a[r] = r < 4 ? 0x22222222 : 0; ones_b[r] = 0x38383838; z[r] = 0; }
    v4f d = mm(ones_a, ones_b, 127, 127);
    f_out[l] = d[0];
    d = mm(ones_a, ones_b, 127, 128 | (129 << 8) | (130 << 16) | (131 << 24));
    f_out[64 + l] = d[0];
    d = mm(ones_a, ones_b, 129, 127);
    f_out[128 + l] = d[0];
    // A impulse
    for (int t = 0; t < 2048; ++t) {
        const int la = t >> 5, na = t & 31;
        v8i a = z;
        if (l == la) a[na >> 3] = 2 << (4 * (na & 7));
        d = mm(a, ones_b, 127, 127);
        int row = -1;
        for (int r = 0; r < 4; ++r) if (d[r] != 0.f) row = 4 * (l >> 4) + r;
        // any lane with nonzero reports (all columns should be nonzero for that row)
        unsigned long long m = __ballot(row >= 0);
        if (m) { int src = __ffsll((long long)m) - 1; int rr = __shfl(row, src, 64); if (l == 0) arow[t] = rr | (__popcll(m) << 8); }
        else if (l == 0) arow[t] = -1;
    }
    for (int t = 0; t < 2048; ++t) {
        const int lb = t >> 5, nb = t & 31;
        v8i b = z;
        if (l == lb) b[nb >> 2] = 0x38 << (8 * (nb & 3));
        d = mm(ones_a, b, 127, 127);
        int col = -1;
        for (int r = 0; r < 4; ++r) if (d[r] != 0.f) col = l & 15;
        unsigned long long m = __ballot(col >= 0);
        if (m) { int src = __ffsll((long long)m) - 1; int cc = __shfl(col, src, 64); if (l == 0) bcol[t] = cc | (__popcll(m) << 8); }
        else if (l == 0) bcol[t] = -1;
    }
    // k matching for the A elements of lanes 0, 16, 32, 48 (rows equal, k-groups differ) -> 128 A elements x 2048 B elements
    for (int ta = 0; ta < 128; ++ta) {
        const int la = 16 * (ta >> 5), na = ta & 31;
        v8i a = z;
        if (l == la) a[na >> 3] = 2 << (4 * (na & 7));
        int cnt = 0;
        for (int t = 0; t < 2048; ++t) {
            const int lb = t >> 5, nb = t & 31;
            v8i b = z;
            if (l == lb) b[nb >> 2] = 0x38 << (8 * (nb & 3));
            d = mm(a, b, 127, 127);
            bool nz = false;
            for (int r = 0; r < 4; ++r) nz |= d[r] != 0.f;
            if (__ballot(nz)) { if (l == 0 && cnt < 16) match[ta * 16 + cnt] = t; ++cnt; }
        }
        if (l == 0) nmatch[ta] = cnt;
    }
}
int main() {
    float* f; int *arow, *bcol, *match, *nmatch;
    hipMalloc(&f, 192 * 4); hipMalloc(&arow, 2048 * 4); hipMalloc(&bcol, 2048 * 4); hipMalloc(&match, 128 * 16 * 4); hipMalloc(&nmatch, 128 * 4);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, f, arow, bcol, match, nmatch);
    float hf[192]; int ha[2048], hb[2048], hm[128 * 16], hn[128];
    hipMemcpy(hf, f, sizeof hf, hipMemcpyDeviceToHost); hipMemcpy(ha, arow, sizeof ha, hipMemcpyDeviceToHost); hipMemcpy(hb, bcol, sizeof hb, hipMemcpyDeviceToHost);
    hipMemcpy(hm, match, sizeof hm, hipMemcpyDeviceToHost); hipMemcpy(hn, nmatch, sizeof hn, hipMemcpyDeviceToHost);
    printf("all-ones D: %g %g %g %g | scale_b bytes test (lanes 0,16,32,48): %g %g %g %g | scale_a=129: %g\n", hf[0], hf[1], hf[16], hf[63], hf[64], hf[64 + 16], hf[64 + 32], hf[64 + 48], hf[128]);
    printf("A impulse rows, lane 0 nibbles 0..31: "); for (int n = 0; n < 32; ++n) printf("%d/%d ", ha[n] & 255, ha[n] >> 8); printf("\n");
    printf("A impulse rows, nibble 0 lanes 0..63: "); for (int l = 0; l < 64; ++l) printf("%d ", ha[l * 32] & 255); printf("\n");
    printf("B impulse cols, lane 0 bytes 0..31: "); for (int n = 0; n < 32; ++n) printf("%d/%d ", hb[n] & 255, hb[n] >> 8); printf("\n");
    printf("B impulse cols, byte 0 lanes 0..63: "); for (int l = 0; l < 64; ++l) printf("%d ", hb[l * 32] & 255); printf("\n");
    for (int ta = 0; ta < 128; ta += 1) {
        if (!(ta % 32 < 10 || ta % 32 > 29)) continue;
        printf("A(lane %d, nib %d) matches %d B elems:", 16 * (ta >> 5), ta & 31, hn[ta]);
        for (int c = 0; c < (hn[ta] < 16 ? hn[ta] : 16); ++c) printf(" (l%d,b%d)", hm[ta * 16 + c] >> 5, hm[ta * 16 + c] & 31);
        printf("\n");
    }
    return 0;
}
