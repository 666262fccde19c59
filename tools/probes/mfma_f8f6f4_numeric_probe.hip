#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
// A: fp4 (cbsz=4) 16 x 128, B: fp8 e4m3 (blgp=0) 128 x 16.  Raw per-lane registers are supplied by the host.
__global__ void k(const int* a_regs, const int* b_regs, const int* sa, const int* sb, float* d) {
    const int l = threadIdx.x;
    v8i a, b;
    for (int r = 0; r < 8; ++r) { a[r] = a_regs[l * 8 + r]; b[r] = b_regs[l * 8 + r]; }
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 0, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
static float fp4v(int n) { static const float t[8] = {0, .5f, 1, 1.5f, 2, 3, 4, 6}; return (n & 8 ? -1.f : 1.f) * t[n & 7]; }
static float fp8v(int b) {  // e4m3fn
    int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -v : v;
}
int main() {
    int ha[64 * 8] = {0}, hb[64 * 8] = {0}, hsa[64], hsb[64];
    float A[16][128], B[128][16];
    srand(1);
    // hypothesis: A lane l: row = l & 15, k = 32*(l>>4) + n (nibble n of the first 128 bits, low nibble first)
    //             B lane l: col = l & 15, k = 32*(l>>4) + n (byte n of the 256 bits)
    for (int l = 0; l < 64; ++l) {
        hsa[l] = 127; hsb[l] = 127 + ((l / SDIV) % 3) * SMUL;          // B scale 1, 2 or 4 per lane
        for (int n = 0; n < 32; ++n) {
            int av = rand() % 4;                        // codes 0..3 only (what X needs)
            ha[l * 8 + n / 8] |= av << (4 * (n % 8));
            A[l & 15][32 * (l >> 4) + n] = fp4v(av);
            int bv = rand() % 120;                      // positive fp8 values
            hb[l * 8 + n / 4] |= bv << (8 * (n % 4));
            B[64 * (n >> 4) + 16 * (l >> 4) + (n & 15)][l & 15] = fp8v(bv) * ldexpf(1.f, hsb[l] - 127);
        }
    }
    int *da, *db, *dsa, *dsb; float* dd;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 64 * 4 * 4);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, da, db, dsa, dsb, dd);
    float hd[256]; hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        int i = 4 * (l >> 4) + r, j = l & 15;            // D layout of 16x16 MFMAs
        double ref = 0; for (int kk = 0; kk < 128; ++kk) ref += (double)A[i][kk] * B[kk][j];
        maxerr = fmax(maxerr, fabs(ref - hd[l * 4 + r])); maxref = fmax(maxref, fabs(ref));
    }
    printf("max |D - ref| = %g (max ref %g)  D[0]=%g\n", maxerr, maxref, hd[0]);
    return 0;
}
