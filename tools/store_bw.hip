// Per-CU global store throughput on MI355X: how many bytes per clock can ONE CU store when few / all CUs are storing?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_bw tools/store_bw.hip && /tmp/store_bw
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(4))) float f32x4;

// every wave writes `iters` rounds of 64 lanes x 16 B; pattern 0: 1 KiB contiguous per wave instruction,
// pattern 1: 16 rows x 64 B (half lines, as one plane of the split-bf16 output layout), pattern 2: 8 rows x 128 B
__global__ __launch_bounds__(512) void store_kernel(f32x4* out, int iters, int pattern, size_t wave_stride) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    char* base = (char*)out + (size_t)wave * wave_stride;
    f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    for (int i = 0; i < iters; ++i) {
        size_t off;
        if (pattern == 0) off = (size_t)i * 1024 + lane * 16;
        else if (pattern == 1) off = (size_t)i * 2048 + (lane >> 2) * 128 + (lane & 3) * 16;
        else off = (size_t)i * 1024 + (lane >> 3) * 128 + (lane & 7) * 16;
        *(f32x4*)(base + off) = v;
    }
}

int main() {
    const size_t bytes = (size_t)4 << 30;
    f32x4* d;
    if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    (void)hipMemset(d, 0, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int pattern = 0; pattern < 3; ++pattern)
        for (int blocks : {8, 32, 64, 128, 256, 512}) {
            const int waves = blocks * 8, iters = 512;
            const size_t per_wave = (size_t)iters * (pattern == 1 ? 2048 : 1024);
            if ((size_t)waves * per_wave > bytes) continue;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(store_kernel, dim3(blocks), dim3(512), 0, 0, d, iters, pattern, per_wave);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            const double written = (double)waves * iters * 1024.0;
            printf("pattern %d  %3d workgroups (8 waves): %.3f ms  %.2f TB/s  %.1f GB/s per workgroup\n", pattern, blocks, best, written / best / 1e9,
                   written / best / 1e6 / blocks);
        }
    return 0;
}
