// Device-side helpers shared by the attention kernels (attention.hip, attention_pipe.hip): LDS tile images and the
// hardware transpose read.  Not part of the C ABI.
#pragma once
#include "common.h"
#include "kernels.h"
#include <type_traits>

namespace cwm {

// Query tile / (batch, head) of work item L (dispatch order) of nqb x nbh items of `rows` query rows each.  Speed only -- the mapping
// is a bijection, any placement is correct:
//  * all query tiles of one (batch, head) go to ONE XCD (workgroups are dealt round-robin over the 8 XCDs by their linear id), so its
//    K / V tiles are fetched into one L2 instead of up to eight (ViT-B/8 encoder: 7 tiles per head, 405 KB of K / V per head)
//  * inside an XCD the ragged last query tile of every head (N = 792: 24 of 128 rows, one active wave) is dispatched after all full
//    tiles, so that those light workgroups fill the tail of the launch instead of being spread through it
__device__ __forceinline__ void attn_tile_of_item(int L, int nqb, int nbh, int nq, int rows, bool remap, int& qt, int& bh) {
    qt = L % nqb;
    bh = L / nqb;
    if (nbh % 8 != 0 || !remap) return;
    const int xcd = L & 7, idx = L >> 3, per = nbh >> 3;
    const int light = (nqb > 1 && (nq - (nqb - 1) * rows) * 2 <= rows) ? 1 : 0;  // last tile at most half full
    const int heavy = nqb - light;
    if (idx < per * heavy) {
        bh = xcd * per + idx / heavy;
        qt = idx - (idx / heavy) * heavy;
    } else {
        bh = xcd * per + (idx - per * heavy);
        qt = nqb - 1;
    }
}
// (the same for a 2-D grid of (nqb, nbh) workgroups: linear id = dispatch order)
__device__ __forceinline__ void attn_tile_of_block(int nq, int rows, bool remap, int& qt, int& bh) {
    attn_tile_of_item(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.y, nq, rows, remap, qt, bh);
}

__device__ __forceinline__ int lds_off128(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// V tile image: key row of 128 bytes, 16-byte chunk c (8 d) stored at c ^ 4 on key rows with bit 1 set
__device__ __forceinline__ int lds_off_v(int row, int chunk) { return row * 128 + ((chunk ^ (((row >> 1) & 1) << 2)) << 4); }

typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x4 lds_read_tr16(const char* ptr) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)ptr);
    return __builtin_bit_cast(bf16x4, v);
}

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
// ds_read_b64_tr_b16 as inline asm (attention.hip / attention_pipe.hip, the P V phase); OFF = immediate byte offset
template <int OFF>
__device__ __forceinline__ u32x2 lds_read_tr16_asm(unsigned addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// all V^T fragments of k-step KS: [d-block][plane][half]
template <int KS, int PLANES>
__device__ __forceinline__ void lds_read_v_step(u32x2 (&dst)[2][PLANES][2], unsigned va0, unsigned va1) {
    constexpr int T = 64 * 64 * 2;
    dst[0][0][0] = lds_read_tr16_asm<KS * 2048>(va0);
    dst[0][0][1] = lds_read_tr16_asm<KS * 2048 + 1024>(va0);
    if constexpr (PLANES == 2) {
        dst[0][PLANES - 1][0] = lds_read_tr16_asm<KS * 2048 + T>(va0);
        dst[0][PLANES - 1][1] = lds_read_tr16_asm<KS * 2048 + T + 1024>(va0);
    }
    dst[1][0][0] = lds_read_tr16_asm<KS * 2048>(va1);
    dst[1][0][1] = lds_read_tr16_asm<KS * 2048 + 1024>(va1);
    if constexpr (PLANES == 2) {
        dst[1][PLANES - 1][0] = lds_read_tr16_asm<KS * 2048 + T>(va1);
        dst[1][PLANES - 1][1] = lds_read_tr16_asm<KS * 2048 + T + 1024>(va1);
    }
}

// s_waitcnt lgkmcnt(N) for fragments requested by lds_read_v_step.  The registers are operands of the wait: hipcc treats the
// output of the asm read as available at once, and without the dependency it is free to schedule a consumer (an MFMA) above
// a bare s_waitcnt statement -- which goes unnoticed as long as the LDS answers quickly (one workgroup per CU) and reads stale
// registers when it does not.
template <int N, int PLANES>
__device__ __forceinline__ void lds_wait_v_step(u32x2 (&v)[2][PLANES][2]) {
    if constexpr (PLANES == 2)
        asm volatile("s_waitcnt lgkmcnt(%8)"
                     : "+v"(v[0][0][0]), "+v"(v[0][0][1]), "+v"(v[0][1][0]), "+v"(v[0][1][1]), "+v"(v[1][0][0]), "+v"(v[1][0][1]), "+v"(v[1][1][0]), "+v"(v[1][1][1])
                     : "n"(N)
                     : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(v[0][0][0]), "+v"(v[0][0][1]), "+v"(v[1][0][0]), "+v"(v[1][0][1]) : "n"(N) : "memory");
}

// max over the lane pair (l, l ^ 32) -- the two key halves of a query in the 32x32 accumulator layout -- with one v_permlane32_swap
// (__shfl_xor would be a ds_bpermute round trip through the LDS queue plus five VALU instructions of index arithmetic)
__device__ __forceinline__ float max_lane_xor32(float x) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}

}  // namespace cwm
