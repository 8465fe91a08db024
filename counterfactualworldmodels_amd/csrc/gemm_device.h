// Device-side building blocks of the GEMM kernels (gemm.hip): LDS tile addressing, exact-erf GELU,
// row maps and the fused epilogues.  Not part of the C ABI.
#pragma once
#include "common.h"
#include "kernels.h"
#include <type_traits>

namespace cwm {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int BK>
__device__ __forceinline__ int lds_swizzle(int row) {
    if constexpr (BK == 64) {
        return (row >> 1) & 7;
    } else {
        // 64-byte rows: 4 rows share one 256-byte bank row; permute so each ds_read_b128 lane group
        // (rows {0-3,12-15} at chunk c, rows {4-11} at chunk c^1) covers 16 distinct 16-byte slots.
        return (0x78 >> (2 * ((row >> 2) & 3))) & 3;  // {0,2,3,1}
    }
}

template <int BK>
__device__ __forceinline__ int lds_off(int row, int chunk) {
    return row * (BK * 2) + ((chunk ^ lds_swizzle<BK>(row)) << 4);
}

// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)) (nn.GELU default, VideoMAE/utils.py:38,49).  erf by Abramowitz & Stegun 7.1.28,
// erf(z) = 1 - (1 + a1 z + ... + a6 z^6)^-16 for z >= 0, with the 1 / sqrt 2 folded into the coefficients and the odd symmetry
// written as GELU(x) = max(x, 0) - 0.5 |x| (1 - erf(|x| / sqrt 2)):  6 fma + 4 squarings + v_rcp + max + mul + fma = 13 full-rate
// instructions and ONE transcendental (7.1.26, used before: 14 + rcp + exp).  |gelu error| <= 7.1e-7 over [-9, 9] in fp32 against
// the float64 erf form (7.1.26: 4.7e-7), i.e. at the level of fp32 rounding of the exact form; branch-free; large |x| overflow
// p^16 to inf and v_rcp(inf) = 0 gives the exact limits x and -0.  The epilogue of fc1 is VALU-bound.
__device__ __forceinline__ float gelu_erf(float x) {
    const float ax = fabsf(x);
    float p = fmaf(5.3829749e-06f, ax, 4.8890637e-05f);  // a6 / 8, a5 / (4 sqrt 2)
    p = fmaf(p, ax, 3.8003574e-05f);                                           // a4 / 4
    p = fmaf(p, ax, 3.2776264e-03f);                                           // a3 / (2 sqrt 2)
    p = fmaf(p, ax, 2.1141006e-02f);                                           // a2 / 2
    p = fmaf(p, ax, 4.9867347e-02f);                                           // a1 / sqrt 2
    p = fmaf(p, ax, 1.0f);
    p *= p;
    p *= p;
    p *= p;
    p *= p;
    const float r = __builtin_amdgcn_rcpf(p);  // 1 - erf(|x| / sqrt 2)
    return fmaf(-0.5f * ax, r, fmaxf(x, 0.0f));
}

// The same GELU on four values at once, written on 2-element vectors so that the compiler can use the packed fp32 VALU instructions
// (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of fma per instruction -- the epilogue runs with the matrix pipe idle, so there is
// nothing for a packed instruction to collide with); bit-identical to four gelu_erf calls.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x4 gelu_erf4(f32x4 v) {
    f32x4 out;
#ifdef CWM_GELU_SCALAR  // A/B builds (CWM_HIPCC_EXTRA=-DCWM_GELU_SCALAR with CWM_HIP_LIB_OUT)
    for (int e = 0; e < 4; ++e) out[e] = gelu_erf(v[e]);
    return out;
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 x = f32x2{v[2 * h], v[2 * h + 1]};
        const f32x2 ax = f32x2{fabsf(x[0]), fabsf(x[1])};
        f32x2 p = __builtin_elementwise_fma(f32x2{5.3829749e-06f, 5.3829749e-06f}, ax, f32x2{4.8890637e-05f, 4.8890637e-05f});
        p = __builtin_elementwise_fma(p, ax, f32x2{3.8003574e-05f, 3.8003574e-05f});
        p = __builtin_elementwise_fma(p, ax, f32x2{3.2776264e-03f, 3.2776264e-03f});
        p = __builtin_elementwise_fma(p, ax, f32x2{2.1141006e-02f, 2.1141006e-02f});
        p = __builtin_elementwise_fma(p, ax, f32x2{4.9867347e-02f, 4.9867347e-02f});
        p = __builtin_elementwise_fma(p, ax, f32x2{1.0f, 1.0f});
        p *= p;
        p *= p;
        p *= p;
        p *= p;
        const f32x2 r = f32x2{__builtin_amdgcn_rcpf(p[0]), __builtin_amdgcn_rcpf(p[1])};
        const f32x2 pos = f32x2{fmaxf(x[0], 0.0f), fmaxf(x[1], 0.0f)};
        const f32x2 res = __builtin_elementwise_fma(ax * f32x2{-0.5f, -0.5f}, r, pos);
        out[2 * h] = res[0];
        out[2 * h + 1] = res[1];
    }
    return out;
}

struct RowMap {
    int out_row, res_row, b, tok;
};

__device__ __forceinline__ RowMap map_row(const GemmParams& p, int m) {
    RowMap r;
    if (p.rows_in > 0) {
        r.b = m / p.rows_in;
        r.tok = m - r.b * p.rows_in;
        r.out_row = r.b * p.rows_out + r.tok + p.out_row_offset;
        r.res_row = p.resid_rowmap ? p.resid_rowmap[r.b * p.map_stride + r.tok] : r.out_row;
    } else {
        r.b = 0;
        r.tok = m;
        r.out_row = m;
        r.res_row = m;
    }
    return r;
}

// Q / K / V output base.  (Written as a select of three loaded VALUES: hipcc turns `which == 0 ? p.q_out : ...` over the
// three adjacent pointer fields into a dynamic index into the kernel-argument struct, which then lives in scratch.)
__device__ __forceinline__ bf16* qkv_out_base(const GemmParams& p, int which) {
    bf16 *qo = p.q_out, *ko = p.k_out, *vo = p.v_out;
    asm volatile("" : "+s"(qo), "+s"(ko), "+s"(vo));
    return which == 0 ? qo : which == 1 ? ko : vo;
}

// One transposed accumulator fragment: v[r] = C[m][nb + ncol + r] (nb = the fragment's first column, wave-uniform;
// ncol = (lane >> 4) * 4), with rm = map_row(p, m).  Shared by every GEMM kernel of this file.
template <int PLANES>
__device__ __forceinline__ void epilogue_frag(const GemmParams& p, const RowMap& rm, int nb, int ncol, f32x4 v) {
    const int n = nb + ncol;
    if (p.debug & 2) {
        asm volatile("" ::"v"(v));
        return;
    }
    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
    if (p.epi == EPI_F32) {
        if (p.resid) v += *reinterpret_cast<const f32x4*>(p.resid + (size_t)rm.res_row * p.ldr + n);
        *reinterpret_cast<f32x4*>(p.C + (size_t)rm.out_row * p.ldc + n) = v;
        return;
    }
    bf16* dst;
    int64_t plane;
    if (p.epi == EPI_QKV) {
        const int D = p.qkv_dim;
        const int which = nb / D;  // 0 q, 1 k, 2 v (uniform per 16-column fragment)
        const int c = n - which * D;
        const int h = c / p.head_dim, d = c - h * p.head_dim;
        if (which == 0) v *= p.q_scale;
        dst = qkv_out_base(p, which) + ((size_t)(rm.b * p.heads + h) * p.n_tok + rm.tok) * p.head_dim + d;
        plane = p.qk_plane;
    } else {
        if (p.epi == EPI_BF16_GELU) {
#pragma unroll
            for (int r = 0; r < 4; r += 4) v = gelu_erf4(v);
        }
        dst = p.out_hi + a_pos<PLANES>(rm.out_row, p.ldo, n);  // A-operand layout of the next GEMM
        plane = kLoOffset;
    }
    bf16x4 hv, lv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bf16 hi = (bf16)v[r];
        hv[r] = hi;
        lv[r] = (bf16)(v[r] - (float)hi);
    }
    if (p.debug & 1) {
        asm volatile("" ::"v"(hv), "v"(lv), "v"(dst));
        return;
    }
    *reinterpret_cast<bf16x4*>(dst) = hv;
    if constexpr (PLANES == 2) *reinterpret_cast<bf16x4*>(dst + plane) = lv;
}

// Transposed accumulators: acc[i][j][r] = C[m0 + wr*64 + i*16 + (lane&15)][n0 + wc*64 + j*16 + (lane>>4)*4 + r]
template <int PLANES, int FM, int FN>
__device__ __forceinline__ void epilogue_rows(const GemmParams& p, const f32x4 (&acc)[FM][FN], int m0, int n0, int wr, int wc, int lane) {
    const int ncol = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wr * (16 * FM) + i * 16 + (lane & 15);
        if (m >= p.M) continue;
        const RowMap rm = map_row(p, m + p.m_offset);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int nb = n0 + wc * (16 * FN) + j * 16;  // fragment's first column (wave-uniform)
            if (nb >= p.N) continue;
            epilogue_frag<PLANES>(p, rm, nb, ncol, acc[i][j]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// LDS-staged epilogue (p.staged): the accumulator layout (16 rows x 4 lane groups of 4 columns) gives 32- / 64-byte
// runs per row and makes every lane redo the row -> (batch, token) division for each of its rows; measured, that
// epilogue was 30 % (parity) to 47 % (fast) of a K = 768 GEMM.  Here the workgroup first writes a row table
// (one division per tile row) to LDS; then every wave pushes its tile through a private 8-KiB LDS buffer one
// 64-row x 32-column piece at a time and reads it back row-major, so that every global access of the epilogue is a
// 16-byte lane access forming full 128-byte (64-byte: fast-mode bf16, Q/K/V) row segments.
//   piece buffer: [64 rows][128 B], 16-byte chunk c of row r at c ^ (r & 7)
//     fp32: chunk = 4 columns;  bf16: chunks 0-3 = hi of 8 columns each, chunks 4-7 = lo (parity only)
//   row table: out_row / residual row (EPI_F32, EPI_BF16*), b * heads * n_tok + token / - (EPI_QKV); -1 = row >= M
__device__ __forceinline__ void epilogue_row_table(const GemmParams& p, int4* tab, int m0, int bm, int tid) {
    if (tid < bm) {
        const int m = m0 + tid;
        int4 e = make_int4(-1, 0, 0, 0);
        if (m < p.M) {
            const RowMap rm = map_row(p, m + p.m_offset);
            if (p.epi == EPI_QKV) e.x = rm.b * p.heads * p.n_tok + rm.tok;
            else { e.x = rm.out_row; e.y = rm.res_row; }
        }
        tab[tid] = e;
    }
}

// Where lanes with nothing to write (rows >= M, columns >= N) send their 16 bytes, and where they read a dummy residual.
// Every global access of the staged epilogue is UNCONDITIONAL: exec-masked `if (valid) store` branches made hipcc lose count
// of the vector-memory queue and wait `vmcnt(0)` before the next piece's first load result -- and since the counter retires in
// order, each such wait also drained every store issued before it (measured: 40-60 us of a 100-us projection).
static __device__ __attribute__((aligned(16))) float g_epilogue_trash[64 * 4];

// A sequence of NPIECE 64-row x 32-column pieces of one wave through its 8-KiB LDS buffer.
//   frag(pi, i, j): piece pi's fragment of rows 16 i .., columns 16 j .. (i < 4, j < 2); row0_of(pi): its first tile row;
//   nb_of(pi): its first column (global, multiple of 32).
// Order of the vector-memory operations: every bias load first; a piece is read back in two halves, and the residual rows of
// the NEXT half are requested BEFORE the stores of this half are issued, so the (in-order) wait for them never waits for a store.
template <int PLANES, int NPIECE, class Frag, class Row0, class Nb>
__device__ __forceinline__ void epilogue_piece_seq(const GemmParams& p, Frag frag, Row0 row0_of, Nb nb_of, char* wlds, const int4* tab, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    const bool f32_out = p.epi == EPI_F32;
    constexpr int NS = 8;  // read-back iterations of 8 rows (128-byte rows); fast-mode bf16 outputs: 4 iterations of 16 rows (64-byte rows)
    const bool wide = f32_out || PLANES == 2;
    // (explicitly GLOBAL pointers: a select between an output address and the trash buffer is otherwise a generic pointer
    // and the accesses become flat_load / flat_store, which also count on lgkmcnt and complete out of order)
    typedef __attribute__((address_space(1))) f32x4 gf32x4;
    gf32x4* const trash = (gf32x4*)(g_epilogue_trash + lane * 4);

    f32x4 bias[NPIECE][2];
#pragma unroll
    for (int pi = 0; pi < NPIECE; ++pi)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = nb_of(pi) + j * 16;
            bias[pi][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias && n < p.N) bias[pi][j] = *reinterpret_cast<const f32x4*>(p.bias + n + fq * 4);
        }

    // residual rows of half a piece (unit u = 2 pi + half: read-back iterations 4 half .. 4 half + 3), EPI_F32 only
    auto load_resid = [&](int u, f32x4 (&rv)[NS / 2]) {
        const int pi = u >> 1, hf = u & 1;
        const int n = nb_of(pi) + (lane & 7) * 4;
#pragma unroll
        for (int s = 0; s < NS / 2; ++s) {
            const int4 info = tab[row0_of(pi) + (hf * 4 + s) * 8 + (lane >> 3)];
            const bool ok = p.resid && info.x >= 0 && n < p.N;
            const gf32x4* src = ok ? (const gf32x4*)(p.resid + (size_t)info.y * p.ldr + n) : trash;
            rv[s] = *src;  // raw: the "valid" select is applied where the value is used, so that nothing waits for the load here
        }
    };
    f32x4 rv_next[NS / 2];
    if (f32_out) load_resid(0, rv_next);

#pragma unroll
    for (int pi = 0; pi < NPIECE; ++pi) {
        const int nb = nb_of(pi), row0 = row0_of(pi);
        int which = 0;
        bf16* qkv_base = nullptr;
        int qh = 0, qd = 0;
        if (p.epi == EPI_QKV) {
            which = nb / p.qkv_dim;
            qkv_base = qkv_out_base(p, which);
            const int cD = nb - which * p.qkv_dim;
            qh = cD / p.head_dim;
            qd = cD - qh * p.head_dim;
        }
        // ---- accumulators (+ bias, activation, split) -> piece buffer ----
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = i * 16 + fr;
                f32x4 v = frag(pi, i, j) + bias[pi][j];
                if (f32_out) {
                    *reinterpret_cast<f32x4*>(wlds + r * 128 + (((j * 4 + fq) ^ (r & 7)) << 4)) = v;
                } else {
                    if (p.epi == EPI_BF16_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; e += 4) v = gelu_erf4(v);
                    } else if (which == 0 && p.epi == EPI_QKV) {
                        v *= p.q_scale;
                    }
                    bf16x4 hv, lv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bf16 hi = (bf16)v[e];
                        hv[e] = hi;
                        lv[e] = (bf16)(v[e] - (float)hi);
                    }
                    const int ch = j * 2 + (fq >> 1), sub = (fq & 1) * 8;
                    *reinterpret_cast<bf16x4*>(wlds + r * 128 + ((ch ^ (r & 7)) << 4) + sub) = hv;
                    if constexpr (PLANES == 2) *reinterpret_cast<bf16x4*>(wlds + r * 128 + (((ch + 4) ^ (r & 7)) << 4) + sub) = lv;
                }
            }
        }
        if (p.debug & 1) continue;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            // ---- half of the piece buffer -> registers (row-major 16-byte chunks), residual added ----
            f32x4 v[NS / 2];
            gf32x4* dst[NS / 2];
            if (wide) {
                const int c = lane & 7;
#pragma unroll
                for (int s = 0; s < NS / 2; ++s) {
                    const int r = (hf * 4 + s) * 8 + (lane >> 3);
                    const int4 info = tab[row0 + r];
                    v[s] = *reinterpret_cast<const f32x4*>(wlds + r * 128 + ((c ^ (r & 7)) << 4));
                    if (f32_out) {
                        const int n = nb + c * 4;
                        const bool ok = info.x >= 0 && n < p.N;
                        if (ok && p.resid) v[s] += rv_next[s];
                        dst[s] = ok ? (gf32x4*)(p.C + (size_t)info.x * p.ldc + n) : trash;
                    } else {
                        const int n = nb + (c & 3) * 8, lo = c >> 2;
                        bf16* d;
                        if (p.epi == EPI_QKV)
                            d = qkv_base + (size_t)lo * p.qk_plane + ((size_t)(info.x + qh * p.n_tok)) * p.head_dim + qd + (c & 3) * 8;
                        else
                            d = p.out_hi + a_pos<2>(info.x, p.ldo, n) + lo * kLoOffset;
                        dst[s] = (info.x >= 0 && n < p.N) ? (gf32x4*)d : trash;
                    }
                }
            } else {
                const int c = lane & 3;
#pragma unroll
                for (int s = 0; s < NS / 4; ++s) {
                    const int r = (hf * 2 + s) * 16 + (lane >> 2);
                    const int4 info = tab[row0 + r];
                    v[s] = *reinterpret_cast<const f32x4*>(wlds + r * 128 + ((c ^ (r & 7)) << 4));
                    const int n = nb + c * 8;
                    bf16* d;
                    if (p.epi == EPI_QKV)
                        d = qkv_base + ((size_t)(info.x + qh * p.n_tok)) * p.head_dim + qd + c * 8;
                    else
                        d = p.out_hi + (size_t)info.x * p.ldo + n;
                    dst[s] = (info.x >= 0 && n < p.N) ? (gf32x4*)d : trash;
                }
            }
            // ---- the next half-piece's residual rows, then this half's stores ----
            if (f32_out && 2 * pi + hf + 1 < 2 * NPIECE) load_resid(2 * pi + hf + 1, rv_next);
            if (wide) {
#pragma unroll
                for (int s = 0; s < NS / 2; ++s) *dst[s] = v[s];
            } else {
#pragma unroll
                for (int s = 0; s < NS / 4; ++s) *dst[s] = v[s];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Direct epilogue (round 4, p.direct): the accumulators go from registers straight to global memory -- no LDS staging buffer, no
// write -> wait -> read-back -> wait chain per piece, so the arithmetic of piece k + 1 (bias, GELU, hi / lo split) issues under the
// stores of piece k.  What made the first per-fragment epilogue slow (8-byte stores, a division per row and fragment) is gone:
//  * rows come from the LDS row table (one division per tile row, epilogue_row_table);
//  * for the bf16 outputs the kernel stages the W tile with its rows PERMUTED inside every 32-row group (w_row_perm below), so that
//    the two 16-column fragments of a lane hold 8 CONSECUTIVE output columns: one 16-byte store per plane and row
//    (column of fragment j, lane group fq, element e:  nb + 8 fq + 4 j + e);  fp32 outputs keep the natural order (nb + 16 j + 4 fq + e),
//    where a lane's 4 columns are 16 bytes already.
// A wave instruction then writes 16 rows x 64 B (lane = 16 fq + fr: row fr, chunk fq).  tools/store_pattern.hip prices that shape at
// 37 GB/s per CU against 100 GB/s for whole 128-byte lines per 8 lanes -- and 22-28 GB/s per CU for EITHER once all 256 CUs store
// at the same time, which is what the epilogue round of a GEMM launch does (profiles/r4_store_pattern.log).
// Same arithmetic per element, in the same order, as the staged form: outputs are bit-identical.
__device__ __forceinline__ int w_row_perm(int row) {  // LDS row 16 j + 4 q + e of a 32-row group holds W row 8 q + 4 j + e
    return (row & ~31) | ((row & 12) << 1) | ((row & 16) >> 2) | (row & 3);
}

template <int PLANES, int NPIECE, class Frag, class Row0, class Nb>
__device__ __forceinline__ void epilogue_direct(const GemmParams& p, Frag frag, Row0 row0_of, Nb nb_of, const int4* tab, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    typedef __attribute__((address_space(1))) f32x4 gf32x4;
    typedef __attribute__((address_space(1))) bf16x8 gbf16x8;
    gf32x4* const trash = (gf32x4*)(g_epilogue_trash + lane * 4);
    const bool f32_out = p.epi == EPI_F32;

    // every bias load first (the vector-memory queue retires in order: nothing below waits for a store because of them)
    f32x4 bias[NPIECE][2];
#pragma unroll
    for (int pi = 0; pi < NPIECE; ++pi)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = nb_of(pi) + (f32_out ? j * 16 + fq * 4 : fq * 8 + j * 4);
            bias[pi][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias && n < p.N) bias[pi][j] = *reinterpret_cast<const f32x4*>(p.bias + n);
        }

    if (f32_out) {
        // residual rows of one piece: requested one piece ahead, BEFORE the stores of the piece in front of them
        auto load_resid = [&](int pi, f32x4 (&rv)[4][2]) {
            const int nb = nb_of(pi), row0 = row0_of(pi);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int4 info = tab[row0 + i * 16 + fr];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = nb + j * 16 + fq * 4;
                    const bool ok = p.resid && info.x >= 0 && n < p.N;
                    const gf32x4* src = ok ? (const gf32x4*)(p.resid + (size_t)max(info.y, 0) * p.ldr + n) : trash;
                    rv[i][j] = *src;
                }
            }
        };
        f32x4 rv[2][4][2];
        load_resid(0, rv[0]);
#pragma unroll
        for (int pi = 0; pi < NPIECE; ++pi) {
            const int nb = nb_of(pi), row0 = row0_of(pi);
            if (pi + 1 < NPIECE) load_resid(pi + 1, rv[(pi + 1) & 1]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int4 info = tab[row0 + i * 16 + fr];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = nb + j * 16 + fq * 4;
                    const bool ok = info.x >= 0 && n < p.N;
                    f32x4 v = frag(pi, i, j) + bias[pi][j];
                    if (ok && p.resid) v += rv[pi & 1][i][j];
                    gf32x4* dst = ok ? (gf32x4*)(p.C + (size_t)max(info.x, 0) * p.ldc + n) : trash;
                    if (!(p.debug & 1)) *dst = v;
                    else asm volatile("" ::"v"(v), "v"(dst));
                }
            }
        }
        return;
    }

#pragma unroll
    for (int pi = 0; pi < NPIECE; ++pi) {
        const int nb = nb_of(pi), row0 = row0_of(pi);
        const int n = nb + fq * 8;  // this lane's 8 consecutive columns
        bf16* base;
        int64_t lo_off;
        int row_mul, which = 0;
        if (p.epi == EPI_QKV) {
            which = nb / p.qkv_dim;
            const int cD = nb - which * p.qkv_dim;
            const int qh = cD / p.head_dim, qd = cD - qh * p.head_dim;
            base = qkv_out_base(p, which) + (size_t)qh * p.n_tok * p.head_dim + qd + fq * 8;
            row_mul = p.head_dim;
            lo_off = p.qk_plane;
        } else {
            base = p.out_hi + a_pos<PLANES>(0, p.ldo, n);
            row_mul = PLANES * p.ldo;
            lo_off = kLoOffset;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int4 info = tab[row0 + i * 16 + fr];
            bf16x8 hv, lv;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 v = frag(pi, i, j) + bias[pi][j];
                if (p.epi == EPI_BF16_GELU) v = gelu_erf4(v);
                else if (which == 0 && p.epi == EPI_QKV) v *= p.q_scale;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bf16 hi = (bf16)v[e];
                    hv[j * 4 + e] = hi;
                    lv[j * 4 + e] = (bf16)(v[e] - (float)hi);
                }
            }
            const bool ok = info.x >= 0 && n < p.N;
            bf16* d = base + (size_t)max(info.x, 0) * row_mul;  // (info.x < 0: a row past M -- the lane stores to the trash buffer instead)
            gbf16x8* dh = ok ? (gbf16x8*)d : (gbf16x8*)trash;
            gbf16x8* dl = ok ? (gbf16x8*)(d + lo_off) : (gbf16x8*)trash;
            if (p.debug & 1) {
                asm volatile("" ::"v"(hv), "v"(lv), "v"(dh), "v"(dl));
                continue;
            }
            *dh = hv;
            if constexpr (PLANES == 2) *dl = lv;
        }
    }
}

// Whole wave tile (FM x FN fragments at tile rows wrow0.., columns ncol0..) through the direct epilogue.
template <int PLANES, int FM, int FN>
__device__ __forceinline__ void epilogue_direct_tile(const GemmParams& p, const f32x4 (&acc)[FM][FN], const int4* tab, int wrow0, int ncol0, int lane) {
    static_assert(FM % 4 == 0 && FN % 2 == 0, "wave tile must be a multiple of the 64x32 piece");
    constexpr int PJ = FN / 2, NPIECE = (FM / 4) * PJ;
    epilogue_direct<PLANES, NPIECE>(
        p, [&](int pi, int i, int j) { return acc[(pi / PJ) * 4 + i][(pi % PJ) * 2 + j]; }, [&](int pi) { return wrow0 + (pi / PJ) * 64; },
        [&](int pi) { return ncol0 + (pi % PJ) * 32; }, tab, lane);
}

// Whole wave tile (FM x FN fragments at tile rows wrow0.., columns ncol0..) through the staged epilogue.
template <int PLANES, int FM, int FN>
__device__ __forceinline__ void epilogue_staged(const GemmParams& p, const f32x4 (&acc)[FM][FN], char* wlds, const int4* tab, int wrow0, int ncol0, int lane) {
    static_assert(FM % 4 == 0 && FN % 2 == 0, "wave tile must be a multiple of the 64x32 piece");
    constexpr int PJ = FN / 2, NPIECE = (FM / 4) * PJ;
    epilogue_piece_seq<PLANES, NPIECE>(
        p, [&](int pi, int i, int j) { return acc[(pi / PJ) * 4 + i][(pi % PJ) * 2 + j]; }, [&](int pi) { return wrow0 + (pi / PJ) * 64; },
        [&](int pi) { return ncol0 + (pi % PJ) * 32; }, wlds, tab, lane);
}

}  // namespace cwm
