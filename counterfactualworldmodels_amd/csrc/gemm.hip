// bf16 / split-bf16 ("bf16x3") MFMA GEMM for CDNA4 (gfx950):  C[M,N] = A[M,K] * W[N,K]^T  (+ epilogue)
//
// Replaces every `F.linear` on the predictor path (reference call sites: VideoMAE/utils.py:48-53
// fc1/fc2, :93 qkv, :119 proj; vmae.py:547 encoder_to_decoder, :251 head; the Conv3d patch embed of
// VideoMAE/utils.py:174-197 expressed as an im2col GEMM).
//
// * 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each 64x64 = 4x4 MFMA 16x16x32 tiles),
//   two workgroups per CU
// * A and W are both K-contiguous, so one lane's MFMA fragment is one 16-byte LDS read
// * K tiles are staged by LDS-DMA (`global_load_lds_dwordx4`, 1 KiB per wave-instruction, no staging
//   VGPRs, no ds_write pass) into a 2-stage LDS ring; tile t+1 is in flight while tile t is multiplied.
//   An LDS-DMA piece lands linearly (lane l -> base + 16 l), so the bank-conflict XOR swizzle is applied
//   to the per-lane SOURCE address and again on the fragment reads (measured SQ_LDS_BANK_CONFLICT = 0)
// * PLANES==2 ("parity" mode): operands are split into bf16 hi + lo and each product is
//   hi*hi + hi*lo + lo*hi, fp32-accumulated -> ~2^-16 relative operand error instead of 2^-9.  hi and lo
//   are stored interleaved per 32-k block ([32 hi | 32 lo] = one 128-byte line, common.h a_pos) so the
//   tile rows are full cache lines in both modes (64-byte half-line DMA requests filled LDS ~1.5x slower)
// * XCD-aware, grouped tile order so that co-resident tiles of one XCD share A panels / W tiles in L2
// * accumulators are kept TRANSPOSED (D^T = W_frag . A_frag^T): a lane then owns 4 consecutive output
//   columns of one row, so every epilogue access is a 16-byte (fp32) / 8-byte (bf16) vector; the V third
//   of the QKV projection uses the plain orientation instead (4 consecutive tokens per lane) because it
//   is stored transposed ([head][d][token]) for the attention kernel
// * fused epilogues: bias, residual(+row map), exact-erf GELU, bf16 hi/lo split, QKV head scatter
#include "common.h"
#include "kernels.h"

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

// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)) (nn.GELU default, VideoMAE/utils.py:38,49).  erf by Abramowitz & Stegun
// 7.1.26 with v_rcp / v_exp: |gelu error| <= 4e-7 over [-8, 8] in fp32 (checked against scipy), i.e. at
// the level of fp32 rounding of the exact form, at a third of the VALU cost of ocml erff (no branches).
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = x * 0.70710678118654752440f;
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    float poly = fmaf(t, 1.061405429f, -1.453152027f);
    poly = fmaf(t, poly, 1.421413741f);
    poly = fmaf(t, poly, -0.284496736f);
    poly = fmaf(t, poly, 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(-az * az * 1.4426950408889634f);
    const float erf_abs = fmaf(-poly, e, 1.0f);
    return 0.5f * x * (1.0f + copysignf(erf_abs, z));
}

struct RowMap {
    int out_row, res_row, b, tok;
};

__device__ __forceinline__ RowMap map_row(const GemmParams& p, int m) {
    RowMap r;
    if (p.rows_in > 0) {
        r.b = m / p.rows_in;
        r.tok = m - r.b * p.rows_in;
        r.out_row = r.b * p.rows_out + r.tok;
        r.res_row = p.resid_rowmap ? p.resid_rowmap[r.b * p.map_stride + r.tok] : r.out_row;
    } else {
        r.b = 0;
        r.tok = m;
        r.out_row = m;
        r.res_row = m;
    }
    return r;
}

// Transposed accumulators: acc[i][j][r] = C[m0 + wr*64 + i*16 + (lane&15)][n0 + wc*64 + j*16 + (lane>>4)*4 + r]
template <int PLANES>
__device__ __forceinline__ void epilogue_rows(const GemmParams& p, const f32x4 (&acc)[4][4], int m0, int n0, int wr, int wc, int lane) {
    const int ncol = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wr * 64 + i * 16 + (lane & 15);
        if (m >= p.M) continue;
        const RowMap rm = map_row(p, m);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = n0 + wc * 64 + j * 16;  // fragment's first column (wave-uniform)
            if (nb >= p.N) continue;
            const int n = nb + ncol;
            f32x4 v = acc[i][j];
            if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
            if (p.epi == EPI_F32) {
                if (p.resid) v += *reinterpret_cast<const f32x4*>(p.resid + (size_t)rm.res_row * p.ldr + n);
                *reinterpret_cast<f32x4*>(p.C + (size_t)rm.out_row * p.ldc + n) = v;
            } else {
                bf16* dst;
                int64_t plane;
                if (p.epi == EPI_QKV) {
                    const int D = p.qkv_dim;
                    const int which = nb / D;  // 0 q, 1 k (v tiles take epilogue_cols)
                    const int c = n - which * D;
                    const int h = c / p.head_dim, d = c - h * p.head_dim;
                    if (which == 0) v *= p.q_scale;
                    dst = (which == 0 ? p.q_out : p.k_out) + ((size_t)(rm.b * p.heads + h) * p.n_tok + rm.tok) * p.head_dim + d;
                    plane = p.qk_plane;
                } else {
                    if (p.epi == EPI_BF16_GELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
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
                *reinterpret_cast<bf16x4*>(dst) = hv;
                if constexpr (PLANES == 2) *reinterpret_cast<bf16x4*>(dst + plane) = lv;
            }
        }
    }
}

// Plain accumulators (V tiles of the QKV projection):
// acc[i][j][r] = C[m0 + wr*64 + i*16 + (lane>>4)*4 + r][n0 + wc*64 + j*16 + (lane&15)]  -> V^T[(b,h)][d][token]
template <int PLANES>
__device__ __forceinline__ void epilogue_cols_vt(const GemmParams& p, const f32x4 (&acc)[4][4], int m0, int n0, int wr, int wc, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int mbase = m0 + wr * 64 + i * 16 + (lane >> 4) * 4;
        if (mbase >= p.M) continue;
        const int b0 = mbase / p.n_tok, tok0 = mbase - b0 * p.n_tok;
        const bool vec = (mbase + 3 < p.M) && (tok0 + 3 < p.n_tok) && ((tok0 & 3) == 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = n0 + wc * 64 + j * 16;
            if (nb >= p.N) continue;
            const int n = nb + (lane & 15);
            const float bias = p.bias ? p.bias[n] : 0.f;
            const int c = n - 2 * p.qkv_dim;
            const int h = c / p.head_dim, d = c - h * p.head_dim;
            bf16x4 hv, lv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[i][j][r] + bias;
                const bf16 hi = (bf16)v;
                hv[r] = hi;
                lv[r] = (bf16)(v - (float)hi);
            }
            if (vec) {
                bf16* dst = p.vt_out + ((size_t)(b0 * p.heads + h) * p.head_dim + d) * p.n_pad + tok0;
                *reinterpret_cast<bf16x4*>(dst) = hv;
                if constexpr (PLANES == 2) *reinterpret_cast<bf16x4*>(dst + p.vt_plane) = lv;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mbase + r;
                    if (m < p.M) {
                        const int b = m / p.n_tok, tok = m - b * p.n_tok;
                        bf16* dst = p.vt_out + ((size_t)(b * p.heads + h) * p.head_dim + d) * p.n_pad + tok;
                        *dst = hv[r];
                        if constexpr (PLANES == 2) dst[p.vt_plane] = lv[r];
                    }
                }
            }
        }
    }
}

template <int PLANES>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmParams p) {
    constexpr int BM = 128, BN = 128;
    // every K tile row is one 128-byte line in LDS and in memory: 64 k of the single plane (fast), or
    // 32 k as [32 hi | 32 lo] (parity, interleaved operand layout: common.h a_pos)
    constexpr int BK = 64 / PLANES;    // logical k per tile
    constexpr int NI = BM / 8 / 4;     // 1-KiB DMA pieces (8 rows each) per wave per operand
    constexpr int TILE_BYTES = BM * 128;
    constexpr int STAGE_BYTES = TILE_BYTES * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // ---- tile selection: XCD chunking + GROUP_M-grouped order --------------------------------
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int ntiles = tiles_m * tiles_n;
    const int id = xcd_remap(blockIdx.x, ntiles);
    constexpr int GROUP_M = 8;
    const int group_sz = GROUP_M * tiles_n;
    const int g = id / group_sz;
    const int first_m = g * GROUP_M;
    const int gm = min(tiles_m - first_m, GROUP_M);
    const int in_g = id - g * group_sz;
    const int tile_m = first_m + (in_g % gm);
    const int tile_n = in_g / gm;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-lane source pointers of this wave's DMA pieces (piece j covers tile rows RPI*j ...) ----
    const bf16* a_src[NI];
    const bf16* w_src[NI];
#pragma unroll
    for (int jj = 0; jj < NI; ++jj) {
        const int row = (wave * NI + jj) * 8 + lane / 8;
        const int logical = (lane % 8) ^ lds_swizzle<64>(row);
        a_src[jj] = p.A + (size_t)min(m0 + row, p.M - 1) * p.lda * PLANES + logical * 8;
        w_src[jj] = p.W + (size_t)(n0 + row) * p.K * PLANES + logical * 8;
    }
    auto issue_tile = [&](int stage, int k0) {
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) {
            const int j = wave * NI + jj;
            char* da = smem + stage * STAGE_BYTES + j * 1024;
            char* dw = smem + stage * STAGE_BYTES + TILE_BYTES + j * 1024;
            __builtin_amdgcn_global_load_lds((gbl_void*)(a_src[jj] + k0), (lds_void*)da, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void*)(w_src[jj] + k0), (lds_void*)dw, 16, 0, 0);
        }
    };

    // the V third of a QKV projection is accumulated in the plain orientation (block-uniform choice)
    const bool plain = (p.epi == EPI_QKV) && (n0 >= 2 * p.qkv_dim);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (row within tile, k-chunk within a 32-wide k-step)
    // 16-byte chunks of a 128-byte row: fast = [k 0..31 | k 32..63] -> two k-steps; parity = [hi | lo] of one k-step
    const int frow = lane & 15, fq = lane >> 4;
    int a_off[4][2], b_off[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            a_off[i][half] = lds_off<64>(wr * 64 + i * 16 + frow, half * 4 + fq);
            b_off[i][half] = TILE_BYTES + lds_off<64>(wc * 64 + i * 16 + frow, half * 4 + fq);
        }

    const int nk = p.K / BK;
    issue_tile(0, 0);
    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        __syncthreads();  // hipcc drains vmcnt before the barrier: tile t has landed; stage cur^1 is free
        if (t + 1 < nk && !(p.ablate & 1)) issue_tile(cur ^ 1, (t + 1) * 64);  // 64 elements = 128 bytes per tile row
        const char* base = smem + cur * STAGE_BYTES;
        if (p.ablate & 4) continue;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 af[PLANES][4], bfr[PLANES][4];
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    af[pl][i] = *reinterpret_cast<const bf16x8*>(base + a_off[i][PLANES == 1 ? kk : pl]);
                    bfr[pl][i] = *reinterpret_cast<const bf16x8*>(base + b_off[i][PLANES == 1 ? kk : pl]);
                }
            if (p.ablate & 2) {
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(af[pl][i]), "v"(bfr[pl][i]));
                continue;
            }
            if (!plain) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (PLANES == 2) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[0][j], af[1][i], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[1][j], af[0][i], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[0][j], af[0][i], acc[i][j], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (PLANES == 2) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][i], bfr[0][j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], bfr[1][j], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], bfr[0][j], acc[i][j], 0, 0, 0);
                    }
            }
        }
    }
    if (!plain)
        epilogue_rows<PLANES>(p, acc, m0, n0, wr, wc, lane);
    else
        epilogue_cols_vt<PLANES>(p, acc, m0, n0, wr, wc, lane);
}

int g_gemm_ablate = 0;

int launch_gemm(const GemmParams& p, int planes, hipStream_t stream) {
    CWM_REQUIRE(planes == 1 || planes == 2, "gemm: planes must be 1 or 2");
    CWM_REQUIRE(p.K % 64 == 0, "gemm: K=%d must be a multiple of 64", p.K);
    CWM_REQUIRE(p.lda % 8 == 0, "gemm: lda=%d must be a multiple of 8", p.lda);
    CWM_REQUIRE(p.M > 0 && p.N > 0, "gemm: empty problem M=%d N=%d", p.M, p.N);
    CWM_REQUIRE(p.N % 16 == 0, "gemm: N=%d must be a multiple of 16", p.N);
    if (p.epi == EPI_F32) {
        CWM_REQUIRE(p.ldc % 4 == 0 && (!p.resid || p.ldr % 4 == 0), "gemm: ldc/ldr must be multiples of 4");
    } else if (p.epi == EPI_QKV) {
        CWM_REQUIRE(p.rows_in == p.n_tok && p.N == 3 * p.qkv_dim && p.head_dim % 4 == 0, "gemm: bad QKV epilogue setup");
        CWM_REQUIRE(p.qkv_dim % 128 == 0, "gemm: QKV epilogue needs the model width (%d) to be a multiple of the 128-column tile", p.qkv_dim);
        CWM_REQUIRE(p.n_pad % 4 == 0, "gemm: n_pad must be a multiple of 4");
    } else {
        CWM_REQUIRE(p.ldo % 4 == 0, "gemm: ldo must be a multiple of 4");
    }
    const int tiles = ((p.M + 127) / 128) * ((p.N + 127) / 128);
    const size_t smem = 65536;
    typedef void (*kern_t)(const GemmParams);
    static const kern_t kerns[2] = {gemm_bf16_kernel<1>, gemm_bf16_kernel<2>};
    static bool attr_done[2] = {false, false};
    kern_t k = kerns[planes - 1];
    if (!attr_done[planes - 1]) {
        CWM_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_done[planes - 1] = true;
    }
    GemmParams pp = p;
    pp.ablate = g_gemm_ablate;
    hipLaunchKernelGGL(k, dim3(tiles), dim3(256), smem, stream, pp);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace cwm
