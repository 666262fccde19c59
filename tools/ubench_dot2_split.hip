// r06: can v_dot2c_f32_bf16 form the remainder of pass 2's hi + lo split of dR?   rem = dR - f32(bf16(dR))  as
//   rem.x = dot2(hp, {-1, 0}, dR.x),  rem.y = dot2(hp, {0, -1}, dR.y)      (hp = v_cvt_pk_bf16_f32(dR.x, dR.y))
// -- two instructions instead of three (v_lshlrev, v_and, v_pk_add_f32).  (1) is the result the exact difference, for every magnitude the
// kernel can see (|dR'| from 1e-19 to 1)?  (2) what does the instruction cost at three waves per SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_dot2_split.hip -o /tmp/ubd && /tmp/ubd
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__global__ void k_check(const float* in, int n, unsigned long long* bad, float* worst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const f32x2_t dR = {in[2 * i], in[2 * i + 1]};
    const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(dR, bf16x2_t));
    const bf16x2_t hb = __builtin_bit_cast(bf16x2_t, hp);
    const bf16x2_t sx = __builtin_bit_cast(bf16x2_t, 0x0000BF80u), sy = __builtin_bit_cast(bf16x2_t, 0xBF800000u);
    const float rx = __builtin_amdgcn_fdot2_f32_bf16(hb, sx, dR.x, false);
    const float ry = __builtin_amdgcn_fdot2_f32_bf16(hb, sy, dR.y, false);
    const f32x2_t rem = dR - (f32x2_t){__uint_as_float(hp << 16), __uint_as_float(hp & 0xFFFF0000u)};
    if (__float_as_uint(rx) != __float_as_uint(rem.x) && !(rx == 0.f && rem.x == 0.f)) { atomicAdd(bad, 1ull); worst[0] = dR.x; worst[1] = rx; worst[2] = rem.x; }
    if (__float_as_uint(ry) != __float_as_uint(rem.y) && !(ry == 0.f && rem.y == 0.f)) { atomicAdd(bad, 1ull); worst[0] = dR.y; worst[1] = ry; worst[2] = rem.y; }
}

#define ITER 4096
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)
#define KERNEL(NAME, BODY)                                                                                     \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters) {                                       \
        float a[8]; unsigned u[8];                                                                             \
        for (int j = 0; j < 8; ++j) { a[j] = threadIdx.x * 0.001f + j + 1.5f; u[j] = 0x3f803f80u + threadIdx.x; } \
        float c1 = 1.0000001f; asm volatile("" : "+v"(c1));                                                    \
        for (int i = 0; i < iters; ++i) { REP32(BODY) }                                                        \
        float acc = 0;                                                                                         \
        for (int j = 0; j < 8; ++j) acc += a[j] + (float)u[j];                                                 \
        out[blockIdx.x * 256 + threadIdx.x] = acc;                                                             \
    }
#define B_DOT2C(j) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[j]) : "v"(u[j]), "v"(c1));
#define B_DOT2(j) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a[j]) : "v"(u[j]), "v"(c1));
#define B_ADD(j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[j]) : "v"(c1));
#define B_SHL(j) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[j]));
KERNEL(k_dot2c, B_DOT2C) KERNEL(k_dot2, B_DOT2) KERNEL(k_add, B_ADD) KERNEL(k_shl, B_SHL)
typedef void (*kern_t)(float*, int);
static double g_ghz = 2.4;
static void run(const char* name, kern_t fn, float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(256 * 3), dim3(256), 0, 0, out, ITER); (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0); hipLaunchKernelGGL(fn, dim3(256 * 3), dim3(256), 0, 0, out, ITER); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-36s wps=3  %7.3f ms  %6.2f cycles per instruction (at %.2f GHz)\n", name, best, best * 1e-3 * g_ghz * 1e9 / ITER / 3 / 32.0, g_ghz);
}
int main() {
    const int n = 1 << 24;
    float* h = (float*)malloc(n * sizeof(float));
    srand(1);
    for (int i = 0; i < n; ++i) {                       // magnitudes 1e-22 .. 4, both signs, plus special values
        const double e = -22.0 + 22.6 * (rand() / (double)RAND_MAX);
        const double m = 1.0 + rand() / (double)RAND_MAX;
        h[i] = (float)((rand() & 1 ? -1.0 : 1.0) * m * pow(10.0, e));
    }
    h[0] = 0.f; h[1] = -0.f; h[2] = 1.f; h[3] = 1.00390625f; h[4] = 1.0039063f; h[5] = 1e-30f; h[6] = 3.0e-38f; h[7] = 1.17549435e-38f;
    float *d, *worst; unsigned long long* bad;
    (void)hipMalloc(&d, n * sizeof(float)); (void)hipMalloc(&bad, 8); (void)hipMalloc(&worst, 16);
    (void)hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice); (void)hipMemset(bad, 0, 8); (void)hipMemset(worst, 0, 16);
    hipLaunchKernelGGL(k_check, dim3(n / 2 / 256), dim3(256), 0, 0, d, n, bad, worst);
    unsigned long long hb; float hw[3];
    (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hw, worst, 12, hipMemcpyDeviceToHost);
    printf("dot2 remainder vs (dR - hi) over %d values, 1e-22 <= |dR| < 4: %llu differ", n, hb);
    if (hb) printf("  (e.g. dR = %.9g: dot2 %.9g, subtraction %.9g)", hw[0], hw[1], hw[2]);
    printf("\n");
    int clk = 0; (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0); g_ghz = clk * 1e-6;
    float* out; (void)hipMalloc(&out, 256 * 3 * 256 * 4);
    run("v_dot2c_f32_bf16 v, v, v", k_dot2c, out); run("v_dot2_f32_bf16 v, v, v, v", k_dot2, out);
    run("v_add_f32 v, v, v", k_add, out); run("v_lshlrev_b32 v, 16, v", k_shl, out);
    return 0;
}
