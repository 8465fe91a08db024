// How much does a 16-byte-per-lane global store cost per CU by the SHAPE of the wave instruction?  (round 4: can the GEMM epilogue
// store straight from the transposed MFMA accumulators -- lane (fr = lane & 15, fq = lane >> 4) owns 16 B of row fr -- instead of
// going through LDS to get row-major lanes?)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_pattern tools/store_pattern.hip && /tmp/store_pattern
// Every wave owns a [64 rows][row_stride] slab per iteration and writes the first 128 B of each row (one [32 hi | 32 lo] line):
//   pattern 0  "staged":   8 instructions, instruction s covers rows 8s..8s+7, lane -> (row 8s + lane/8, chunk lane%8)        (full lines per 8 lanes)
//   pattern 1  "direct":   8 instructions (i = 0..3 x hi/lo), lane -> (row 16i + lane%16, chunk (lane/16) + 4*half)            (row-fastest lanes, half lines)
//   pattern 2  "direct-q": the same bytes, lane -> (row 16i + lane/4, chunk lane%4 + 4*half)                                 (chunk-fastest lanes, half lines)
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(512) void store_kernel(char* out, int iters, int pattern, size_t row_stride, size_t wave_stride, int slabs) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    char* base = out + (size_t)wave * wave_stride;
    f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    for (int it = 0; it < iters; ++it) {
        char* slab = base + (size_t)(it % slabs) * 64 * row_stride;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            size_t off;
            if (pattern == 0) off = (size_t)(8 * s + (lane >> 3)) * row_stride + (lane & 7) * 16;
            else if (pattern == 1) off = (size_t)(16 * (s >> 1) + (lane & 15)) * row_stride + ((lane >> 4) + 4 * (s & 1)) * 16;
            else off = (size_t)(16 * (s >> 1) + (lane >> 2)) * row_stride + ((lane & 3) + 4 * (s & 1)) * 16;
            *(f32x4*)(slab + off) = v;
        }
    }
}

int main() {
    const size_t bytes = (size_t)8 << 30;
    char* d;
    if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    (void)hipMemset(d, 0, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (size_t row_stride : {(size_t)128, (size_t)3072, (size_t)12288})
        for (int pattern = 0; pattern < 3; ++pattern)
            for (int blocks : {32, 64, 256, 512}) {
                const int waves = blocks * 8;
                const int iters = 128;
                int slabs = (int)(bytes / waves / (64 * row_stride));  // slabs of a wave's region; the iterations wrap around in it
                if (slabs > iters) slabs = iters;
                const size_t per_wave = (size_t)slabs * 64 * row_stride;
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    (void)hipEventRecord(e0);
                    hipLaunchKernelGGL(store_kernel, dim3(blocks), dim3(512), 0, 0, d, iters, pattern, row_stride, per_wave, slabs);
                    (void)hipEventRecord(e1);
                    (void)hipEventSynchronize(e1);
                    float ms;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    best = ms < best ? ms : best;
                }
                const double written = (double)waves * iters * 8192.0;
                printf("row stride %5zu  pattern %d  %3d workgroups (8 waves, %3d slabs): %.3f ms  %.2f TB/s  %.1f GB/s per workgroup\n", row_stride, pattern, blocks,
                       slabs, best, written / best / 1e9, written / best / 1e6 / blocks);
            }
    return 0;
}
