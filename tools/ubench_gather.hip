// r06: what a row gather out of a 12.5 GB resident matrix can deliver, by piece size -- the roof pass 1 (first reader of a fresh batch) sits under.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_gather.hip -o /tmp/ubg && /tmp/ubg        (-> profiles/r06_ubench_gather.txt)
// 800 random rows of 100k x 125056 B; a block owns PB contiguous bytes of every row of its batch split (16 B per lane, 4096 / PB rows per
// load instruction, U instructions in flight); every timed launch reads another batch (16 of them in rotation: 1.6 GB > the 256 MB memory-side cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int PB, int U>
__global__ __launch_bounds__(256) void gather(const uint8_t* __restrict__ xp, int64_t ld, const int* __restrict__ idx, int b, int splits, unsigned* out) {
    constexpr int LPP = PB / 16, RPI = 256 / LPP;                 // lanes per piece, rows per load instruction
    const int chunk = blockIdx.x / splits, split = blockIdx.x % splits;
    const int r0 = (int)((int64_t)b * split / splits), r1 = (int)((int64_t)b * (split + 1) / splits);
    const int lane_col = (threadIdx.x % LPP) * 16, lane_row = threadIdx.x / LPP;
    const int64_t col = (int64_t)chunk * PB + lane_col;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int r = r0; r < r1; r += RPI * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int rr = r + u * RPI + lane_row;
            rr = rr < r1 ? rr : r1 - 1;
            v[u] = *reinterpret_cast<const uint4*>(xp + (int64_t)idx[rr] * ld + col);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x] = 1;
}

// pass 1's own pattern: block = 8 waves, chunk = 512 B of every row; wave w, lane (i = l & 15, q = l >> 4) loads bytes 64 w + 16 q of row i of
// a 16-row tile (a wave instruction = 16 rows x 64 B: half lines, the other half asked for by the neighbouring wave), U tiles in flight
template <int U>
__global__ __launch_bounds__(512) void gather_p1(const uint8_t* __restrict__ xp, int64_t ld, const int* __restrict__ idx, int b, int splits, unsigned* out) {
    const int chunk = blockIdx.x / splits, split = blockIdx.x % splits;
    const int tiles = (b + 15) / 16, t0 = tiles * split / splits, t1 = tiles * (split + 1) / splits;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
    const int64_t col = (int64_t)chunk * 512 + 64 * wave + 16 * q;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int t = t0; t < t1; t += U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int rr = (t + u < t1 ? t + u : t1 - 1) * 16 + i;
            rr = rr < b ? rr : b - 1;
            v[u] = *reinterpret_cast<const uint4*>(xp + (int64_t)idx[rr] * ld + col);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x] = 1;
}
template <int U>
void run_p1(const uint8_t* xp, int64_t ld, int64_t used, const int* idx, int b, int nsets, unsigned* out, int splits) {
    const int chunks = (int)(used / 512);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) gather_p1<U><<<chunks * splits, 512>>>(xp, ld, idx + (w % nsets) * b, b, splits, out);
    float best = 1e9f, sum = 0; const int reps = 32;
    for (int it = 0; it < reps; ++it) {
        CK(hipEventRecord(e0));
        gather_p1<U><<<chunks * splits, 512>>>(xp, ld, idx + ((it + 3) % nsets) * b, b, splits, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms;
    }
    const double bytes = (double)b * chunks * 512;
    printf("pass 1's pattern (16 rows x 64 B per wave instruction), %d tiles in flight, blocks %d (splits %d):  avg %6.1f us  best %6.1f us  = %5.2f / %5.2f TB/s\n",
           U, chunks * splits, splits, sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12);
}

template <int PB, int U>
void run(const uint8_t* xp, int64_t ld, int64_t used, const int* idx, int b, int nsets, unsigned* out, int cus) {
    const int chunks = (int)(used / PB);
    int splits = 1;
    while ((int64_t)chunks * splits < 2 * cus) splits *= 2;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) gather<PB, U><<<chunks * splits, 256>>>(xp, ld, idx + (w % nsets) * b, b, splits, out);
    float best = 1e9f, sum = 0; const int reps = 32;
    for (int it = 0; it < reps; ++it) {
        CK(hipEventRecord(e0));
        gather<PB, U><<<chunks * splits, 256>>>(xp, ld, idx + ((it + 3) % nsets) * b, b, splits, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms;
    }
    const double bytes = (double)b * chunks * PB;
    printf("piece %5d B  in flight %2d x %3d rows  blocks %6d (splits %2d):  avg %6.1f us  best %6.1f us  = %5.2f / %5.2f TB/s\n", PB, U, 4096 / PB,
           chunks * splits, splits, sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12);
}

__global__ void fill(uint4* p, int64_t n) { for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = make_uint4((unsigned)i, (unsigned)(i >> 7), 0x9e3779b9u * (unsigned)i, 1); }
__global__ void stream(const uint4* p, int64_t n, unsigned* out) { uint4 a = make_uint4(0,0,0,0); for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { uint4 v = p[i]; a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w; } if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345678u) out[0] = 1; }

int main() {
    const int64_t rows = 100000, used = 125000 / 2048 * 2048, ld = 125056;       // 61 x 2048 B of every row are read
    const int b = 800, nsets = 16;
    uint8_t* xp; CK(hipMalloc(&xp, rows * ld));
    fill<<<4096, 256>>>((uint4*)xp, rows * ld / 16);
    std::vector<int> h(b * nsets); srand(7);
    for (auto& v : h) v = (int)(((int64_t)rand() * 32768 + rand()) % rows);
    int* idx; CK(hipMalloc(&idx, h.size() * 4)); CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    unsigned* out; CK(hipMalloc(&out, 1 << 22));
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    CK(hipDeviceSynchronize());
    {   // streaming read of 1 GiB for scale
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int64_t n = (1ll << 30) / 16; float best = 1e9f;
        for (int it = 0; it < 6; ++it) { CK(hipEventRecord(e0)); stream<<<cus * 8, 256>>>((const uint4*)xp + (int64_t)it * n, n, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it && ms < best) best = ms; }
        printf("streaming read of 1 GiB: %.2f TB/s\n", (double)(1ll << 30) / (best * 1e-3) / 1e12);
    }
    printf("gather of %d rows x %lld B out of %lld rows x %lld B (%.1f GB), a fresh batch per launch:\n", b, (long long)used, (long long)rows, (long long)ld, rows * ld / 1e9);
    run_p1<3>(xp, ld, used, idx, b, nsets, out, 2);
    run_p1<3>(xp, ld, used, idx, b, nsets, out, 4);
    run_p1<6>(xp, ld, used, idx, b, nsets, out, 2);
    run_p1<12>(xp, ld, used, idx, b, nsets, out, 2);
    run<64, 4>(xp, ld, used, idx, b, nsets, out, cus);
    run<64, 8>(xp, ld, used, idx, b, nsets, out, cus);
    run<128, 4>(xp, ld, used, idx, b, nsets, out, cus);
    run<128, 8>(xp, ld, used, idx, b, nsets, out, cus);
    run<256, 4>(xp, ld, used, idx, b, nsets, out, cus);
    run<256, 8>(xp, ld, used, idx, b, nsets, out, cus);
    run<512, 4>(xp, ld, used, idx, b, nsets, out, cus);
    run<512, 8>(xp, ld, used, idx, b, nsets, out, cus);
    run<512, 16>(xp, ld, used, idx, b, nsets, out, cus);
    run<1024, 8>(xp, ld, used, idx, b, nsets, out, cus);
    run<1024, 16>(xp, ld, used, idx, b, nsets, out, cus);
    run<2048, 8>(xp, ld, used, idx, b, nsets, out, cus);
    run<2048, 16>(xp, ld, used, idx, b, nsets, out, cus);
    run<4096, 16>(xp, ld, used, idx, b, nsets, out, cus);
    return 0;
}
