// r03: what do rocprofv3's FETCH_SIZE / WRITE_SIZE count for the access patterns of pass 2?  (MI355X_MICROARCH.md, HBM: on gfx950
// a wide coalesced stream is tallied at half its bytes; other widths are uncalibrated -- "calibrate on a known byte count in
// your own access pattern".)  Kernels with a KNOWN byte count each, run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
// (tools/pmc_profile.py calib):
//   stream_read     16 B / lane, consecutive lanes -> consecutive addresses (P, m, v of pass 2; the guide's reference pattern)
//   rows64_read     pass 2's X loader: a block reads one 64-byte piece (4 lanes x 16 B) of each of `b` gathered rows, rows a
//                   permutation of a `rows`-row matrix with row stride ld (the byte column of the block)
//   stream_write    16 B / lane consecutive (P, m, v, dP rows)
//   rows64_write    the by-product copy of the batch, tiled by pass 3's chunks: 64-byte pieces at a stride of 128 bytes inside a
//                   [b rows][128 bytes] tile (the block with the other half of the tile runs elsewhere, at another time)
//   slab_write      the dQ slab: 2 KB contiguous per tile and block ([64 samples x 8] floats), blocks 25.6 KB apart
// Every buffer is larger than the 256 MB Infinity Cache or touched once, sizes as in the bench workload (b = 800, M = 500k).
//   hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o /tmp/calib && /tmp/calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

__global__ __launch_bounds__(256) void stream_read(const uint4* __restrict__ src, int64_t n16, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) { const uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void rows64_read(const uint8_t* __restrict__ xp, int64_t ld, const int32_t* __restrict__ idx, int b, uint32_t* __restrict__ sink) {
    const int tid = threadIdx.x, pr = tid >> 2, pc = tid & 3;
    const int64_t off = (int64_t)blockIdx.x * 64 + pc * 16;
    uint32_t acc = 0;
    for (int i0 = 0; i0 < b; i0 += 64) {
        const int r = i0 + pr < b ? i0 + pr : b - 1;
        const uint4 v = *reinterpret_cast<const uint4*>(xp + (int64_t)idx[r] * ld + off);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void stream_write(uint4* __restrict__ dst, int64_t n16) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) dst[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ __launch_bounds__(256) void rows64_write(uint8_t* __restrict__ xg, int64_t ld, int b) {
    const int tid = threadIdx.x, pr = tid >> 2, pc = tid & 3;
    const int64_t off = (int64_t)blockIdx.x * 64 + pc * 16;
    for (int i0 = 0; i0 < b; i0 += 64)
        if (i0 + pr < b) *reinterpret_cast<uint4*>(xg + (off / 128) * ((int64_t)b * 128) + (int64_t)(i0 + pr) * 128 + off % 128) = make_uint4(tid, i0, 2, 3);
}
__global__ __launch_bounds__(256) void slab_write(float4* __restrict__ slab, int b) {      // [chunk][b][8] floats, 128 float4 per 64-sample tile
    for (int i0 = 0; i0 < b; i0 += 64) {
        const int nt = min(64, b - i0);
        for (int e4 = threadIdx.x; e4 < nt * 2; e4 += 256) slab[((int64_t)blockIdx.x * b + i0) * 2 + e4] = make_float4(e4, 1.f, 2.f, 3.f);
    }
}

int main() {
    const int b = 800; const int64_t M = 500000, ld = 125008, rows = 100000, chunks = (M + 255) / 256;
    uint8_t* xp; uint4* big; uint32_t* sink; int32_t* idx; uint8_t* xg; float4* slab;
    (void)hipMalloc(&xp, rows * ld); (void)hipMalloc(&big, 512ll << 20); (void)hipMalloc(&sink, 64); (void)hipMalloc(&idx, b * 4);
    (void)hipMalloc(&xg, (int64_t)b * ld); (void)hipMalloc(&slab, chunks * b * 32);
    (void)hipMemset(xp, 1, rows * ld); (void)hipMemset(big, 1, 512ll << 20);
    std::vector<int32_t> perm(rows); std::iota(perm.begin(), perm.end(), 0); std::mt19937 g(1); std::shuffle(perm.begin(), perm.end(), g);
    (void)hipMemcpy(idx, perm.data(), b * 4, hipMemcpyHostToDevice);
    (void)hipDeviceSynchronize();
    const int64_t n16 = (512ll << 20) / 16;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream_read, dim3(4096), dim3(256), 0, 0, big, n16, sink);
        (void)hipMemcpy(idx, perm.data() + 1000 * (rep + 1), b * 4, hipMemcpyHostToDevice);      // other rows every time: nothing cached
        hipLaunchKernelGGL(rows64_read, dim3((unsigned)chunks), dim3(256), 0, 0, xp, ld, idx, b, sink);
        hipLaunchKernelGGL(stream_write, dim3(4096), dim3(256), 0, 0, big, n16);
        hipLaunchKernelGGL(rows64_write, dim3((unsigned)chunks), dim3(256), 0, 0, xg, ld, b);
        hipLaunchKernelGGL(slab_write, dim3((unsigned)chunks), dim3(256), 0, 0, slab, b);
        (void)hipDeviceSynchronize();
    }
    printf("known_bytes stream_read %lld rows64_read %lld stream_write %lld rows64_write %lld slab_write %lld\n", (long long)(512ll << 20),
           (long long)(chunks * 64 * b), (long long)(512ll << 20), (long long)(chunks * 64 * b), (long long)(chunks * b * 32));
    return 0;
}
