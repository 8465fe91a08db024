// Shared device/host helpers for the CDNA4 (gfx950) VMAE predictor kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cwm {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
// NB: use this native vector (not HIP's struct uint4) for register-staged tiles: arrays of the HIP
// struct types are not promoted to registers by hipcc and end up in scratch.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int kWave = 64;

// Split an fp32 value into bf16 hi + bf16 lo (hi + lo carries ~16 significand bits).
__device__ __forceinline__ void split_bf16(float v, bf16& hi, bf16& lo) {
    hi = (bf16)v;
    lo = (bf16)(v - (float)hi);
}

// GEMM A/W operand layout.  PLANES == 1 ("fast"): plain [rows][ld] bf16.  PLANES == 2 ("parity"): ONE buffer
// [rows][2*ld] with the hi and lo planes interleaved per 32-element k block -- [32 hi | 32 lo] = one 128-byte
// line -- so that the GEMM's LDS-DMA requests are full cache lines in both modes (64-byte half-line requests
// measured ~1.5x slower per byte).  a_pos() is the position of the hi element; the lo element is 32 further.
template <int PLANES>
__device__ __forceinline__ size_t a_pos(int64_t row, int ld, int col) {
    if constexpr (PLANES == 1) return (size_t)row * ld + col;
    else return (size_t)row * 2 * ld + ((col >> 5) << 6) + (col & 31);
}
constexpr int kLoOffset = 32;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// The same butterfly without the LDS: DPP inside a row of 16 lanes (xor 1, xor 2, then mirrors of quad- / half-row-uniform values),
// v_permlane16_swap / v_permlane32_swap across rows (gfx950).  __shfl_xor is a ds_bpermute round trip per step -- six dependent LDS
// latencies per reduction.  Every lane gets the sum; the order of the additions differs from wave_sum (fp32 rounding).
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));  // row_mirror
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Sum over each aligned group of 8 consecutive lanes (every lane of the group gets the total): DPP only, no LDS round trip.
__device__ __forceinline__ float group8_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    return v;
}

// XCD-aware bijective remap of a linear workgroup id (guide §5.5 T1): blocks b and b+8 share an
// XCD under round-robin dispatch, so give every XCD a contiguous chunk of the logical tile order.
// Speed only; any placement is correct.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    int q = nwg / nx, r = nwg % nx;
    int xcd = bid % nx, idx = bid / nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace cwm

// ---- host-side error plumbing (C ABI never throws) ------------------------------------------
void cwm_set_error(const char* fmt, ...);
// hipFuncAttributeMaxDynamicSharedMemorySize for a kernel, once per (device, kernel): the attribute is per device, so a process-wide
// "done" flag would leave a second device without it (engine.hip; thread-safe).  Returns 0 or CWM_ERR_HIP.
int cwm_set_max_lds(const void* kernel, int bytes);

// A handle (model, communicator) belongs to the HIP device that was current when it was created: its weights, workspace, streams and RCCL
// communicator live there, and a launch from another current device would run on the wrong GPU without any error.  0 when the calling thread's
// current device is `handle_device`, else CWM_ERR_INVALID with a message naming `what` (engine.hip).
int cwm_require_device(int handle_device, const char* what);

#define CWM_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            cwm_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return -2;                                                                       \
        }                                                                                    \
    } while (0)

#define CWM_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            cwm_set_error(__VA_ARGS__);   \
            return -1;                    \
        }                                 \
    } while (0)
