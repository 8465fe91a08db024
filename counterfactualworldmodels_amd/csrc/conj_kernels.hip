// Kernels specific to the IMU-conditioned conjoined padded predictor (BASELINE configs[4]):
// null-token padding bookkeeping, IMU tubelet gather, short-sequence self-attention for the context
// stream (head_dim 32, <= 64 tokens) and the bidirectional cross attention between the N-token RGB
// stream and the M <= 64-token IMU stream.  The cross/small attentions here are the exact-fp32 VALU forms: the model runs the MFMA
// kernels of conj_attention.hip where the shapes allow (every shape of the imu400 model) and keeps these for other head widths and as
// the reference of the A/B test (debug switch "conj_attn" = 0).
#include "common.h"
#include "kernels.h"

namespace cwm {

// ---------------------------------------------------------------------------------------------
// `_set_padding_mask` (conjoined_vmae.py:49-116): pad slot j of row b is VISIBLE iff j < vmax - visible(b)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pad_mask_kernel(const uint8_t* mask, int N, int P, int vmax, uint8_t* ext) {
    __shared__ int cnt[256];
    const int b = blockIdx.x, t = threadIdx.x;
    const uint8_t* m = mask + (size_t)b * N;
    uint8_t* e = ext + (size_t)b * (N + P);
    int c = 0;
    for (int i = t; i < N; i += 256) {
        const uint8_t v = m[i] ? 1 : 0;
        e[i] = v;
        c += (v == 0);
    }
    cnt[t] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) cnt[t] += cnt[t + s];
        __syncthreads();
    }
    const int vis = cnt[0];
    for (int j = t; j < P; j += 256) e[N + j] = (j < vmax - vis) ? 0 : 1;
}

int launch_pad_mask(const uint8_t* mask, int B, int N, int P, int vmax, uint8_t* ext_mask, hipStream_t stream) {
    hipLaunchKernelGGL(pad_mask_kernel, dim3(B), dim3(256), 0, stream, mask, N, P, vmax, ext_mask);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void fix_pad_rows_kernel(float* x, const int* perm, int perm_stride, int n_rows, int n_real, int D, const float* token, int64_t total) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int64_t row = gid / D;
    const int d = (int)(gid - row * D);
    const int b = (int)(row / n_rows), i = (int)(row - (int64_t)b * n_rows);
    if (perm[(size_t)b * perm_stride + i] >= n_real) x[gid] = token[d];
}

int launch_fix_pad_rows(float* x, const int* perm, int B, int perm_stride, int n_rows, int n_real, int D, const float* token, hipStream_t stream) {
    const int64_t total = (int64_t)B * n_rows * D;
    hipLaunchKernelGGL(fix_pad_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, perm, perm_stride, n_rows, n_real, D, token, total);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void zero_pad_out_rows_kernel(float* y, const int* perm, int perm_stride, int n_vis, int n_out, int n_real, int D, int64_t total) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int64_t row = gid / D;
    const int b = (int)(row / n_out), j = (int)(row - (int64_t)b * n_out);
    if (perm[(size_t)b * perm_stride + n_vis + j] >= n_real) y[gid] = 0.f;
}

int launch_zero_pad_out_rows(float* y, const int* perm, int B, int perm_stride, int n_vis, int n_out, int n_real, int D, hipStream_t stream) {
    const int64_t total = (int64_t)B * n_out * D;
    hipLaunchKernelGGL(zero_pad_out_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, y, perm, perm_stride, n_vis, n_out, n_real, D, total);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// IMU tubelet gather: `IMU` preprocessor (preprocessor.py:199-206) + Conv3d(6 -> 384, k = (16,1,1))
// im2col (conjoined_vmae.py:1110-1125): token l, K index c*16 + s  <-  imu[b][c][16 l + s]
// ---------------------------------------------------------------------------------------------
template <int PLANES>
__global__ void imu_gather_kernel(const ImuGatherParams p) {
    const int64_t total = (int64_t)p.B * p.n_rows * p.ld;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int64_t row = gid / p.ld;
    const int k = (int)(gid - row * p.ld);
    const int b = (int)(row / p.n_rows), i = (int)(row - (int64_t)b * p.n_rows);
    const int tau = p.perm[(size_t)b * p.perm_stride + i];
    float v = 0.f;
    if (tau < p.n_real && k < p.C * p.tubelet) {
        const int c = k / p.tubelet, s = k - c * p.tubelet;
        v = p.imu[((size_t)b * p.C + c) * p.L + tau * p.tubelet + s];
    }
    bf16 hi, lo;
    split_bf16(v, hi, lo);
    bf16* dst = p.out + a_pos<PLANES>(row, p.ld, k);
    *dst = hi;
    if constexpr (PLANES == 2) dst[kLoOffset] = lo;
}

int launch_imu_gather(const ImuGatherParams& p, int planes, hipStream_t stream) {
    const int64_t total = (int64_t)p.B * p.n_rows * p.ld;
    if (planes == 1)
        hipLaunchKernelGGL(imu_gather_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL(imu_gather_kernel<2>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Short-sequence self-attention (context stream): one 64-thread workgroup per (batch, head), thread n
// owns query n; K and V of the head live in LDS.  `Attention.forward` VideoMAE/utils.py:87-121, fp32.
// ---------------------------------------------------------------------------------------------
template <int PLANES>
__global__ __launch_bounds__(64) void small_attention_kernel(const SmallAttnParams p) {
    constexpr int MAXN = 64, MAXD = 64;
    __shared__ float ks[MAXN][MAXD + 1], vs[MAXN][MAXD + 1];
    const int bh = blockIdx.x, b = bh / p.heads, h = bh - b * p.heads;
    const int hd = p.head_dim, D = p.heads * hd, N = p.n_tok;
    const int t = threadIdx.x;
    const float* base = p.qkv + (size_t)b * N * 3 * D + h * hd;
    for (int i = t; i < N * hd; i += 64) {
        const int n = i / hd, d = i - n * hd;
        ks[n][d] = base[(size_t)n * 3 * D + D + d];
        vs[n][d] = base[(size_t)n * 3 * D + 2 * D + d];
    }
    __syncthreads();
    if (t >= N) return;
    float q[MAXD];
    const float scale = 1.0f / sqrtf((float)hd);
#pragma unroll
    for (int d = 0; d < MAXD; ++d) q[d] = d < hd ? base[(size_t)t * 3 * D + d] * scale : 0.f;
    float mx = -1e30f, l = 0.f;
    float o[MAXD];
#pragma unroll
    for (int d = 0; d < MAXD; ++d) o[d] = 0.f;
#pragma unroll 1
    for (int m = 0; m < N; ++m) {  // online softmax: no per-thread score array (it would live in scratch)
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < MAXD; ++d)
            if (d < hd) a = fmaf(q[d], ks[m][d], a);
        const float mn = fmaxf(mx, a);
        const float alpha = expf(mx - mn), pm = expf(a - mn);
        mx = mn;
        l = l * alpha + pm;
#pragma unroll
        for (int d = 0; d < MAXD; ++d)
            if (d < hd) o[d] = fmaf(pm, vs[m][d], o[d] * alpha);
    }
    const float inv = 1.0f / l;
#pragma unroll
    for (int d = 0; d < MAXD; ++d)
        if (d < hd) {
            bf16 hi, lo;
            split_bf16(o[d] * inv, hi, lo);
            bf16* dst = p.o + a_pos<PLANES>((int64_t)b * N + t, p.ldo, h * hd + d);
            *dst = hi;
            if constexpr (PLANES == 2) dst[kLoOffset] = lo;
        }
}

int launch_small_attention(const SmallAttnParams& p, int planes, hipStream_t stream) {
    CWM_REQUIRE(p.n_tok > 0 && p.n_tok <= 64 && p.head_dim > 0 && p.head_dim <= 64, "small_attention: needs n_tok <= 64 and head_dim <= 64 (got %d, %d)", p.n_tok, p.head_dim);
    if (planes == 1)
        hipLaunchKernelGGL(small_attention_kernel<1>, dim3(p.B * p.heads), dim3(64), 0, stream, p);
    else
        hipLaunchKernelGGL(small_attention_kernel<2>, dim3(p.B * p.heads), dim3(64), 0, stream, p);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Bidirectional cross attention (`BidirectionalCrossAttention.forward`, transformer.py:314-378,
// shared_similarity=False).  Per head h the 2*hd-wide slice of qk / qk_src splits into
//   [0,hd):   attn   = softmax_M( scale * qk1 . qk_src1^T )   -> y     = attn   . v_src   (main update)
//   [hd,2hd): attn_s = softmax_N( scale * qk_src2 . qk2^T )   -> y_src = attn_s . v       (context update)
// N = 3140 .. 6336 main tokens against M = 25 / 50 context tokens: 4 GFLOP per call, i.e. nothing -- the first version of these
// kernels (one workgroup per 4 main tokens, each re-loading the whole context; one workgroup per context token, each
// re-reading all of v) took 5 ms per call = a third of the IMU-conditioned forward.  Now:
//  A  cross_attn_main_kernel   64 main tokens per workgroup (lane = token), context K1 / K2 / V_src of the (b, h) pair in LDS and
//     read as wave-wide broadcasts; wave w computes the scores of context tokens m = w, w + 4, ...; main-side softmax and
//     P . V_src per 32-column group, written as whole [32 hi | 32 lo] operand blocks; the src-side scores go to scores_t[b,h,m,n].
//  B  cross_attn_src_partial_kernel   (b, h) x 16 splits of the main tokens: local max per m, then p = exp(s - max) staged per 64
//     tokens in LDS and thread d accumulating acc[m] += p[m][n] v[n][d] for ALL m at once (v is read once per split).
//  C  cross_attn_src_combine_kernel   merges the 16 (max, sum, acc) partials per (b, h, m) and writes y_src.
// ---------------------------------------------------------------------------------------------
constexpr int kCrossSplit = 16;

template <int PLANES>
__global__ __launch_bounds__(256) void cross_attn_main_kernel(const CrossAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int hd = p.head_dim, M = p.M, D = p.heads * hd, N = p.N;
    float* k1 = lds;                    // [M][hd]
    float* k2 = k1 + (size_t)M * hd;    // [M][hd]
    float* vsrc = k2 + (size_t)M * hd;  // [M][hd]
    float* s1 = vsrc + (size_t)M * hd;  // [64][M + 1] main-side scores of this workgroup's tokens
    const int bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads;
    for (int i = threadIdx.x; i < M * hd; i += 256) {
        const int m = i / hd, d = i - m * hd;
        const float* r = p.qk_src + ((size_t)b * M + m) * 2 * D + h * 2 * hd;
        k1[i] = r[d];
        k2[i] = r[hd + d];
        vsrc[i] = p.v_src[((size_t)b * M + m) * D + h * hd + d];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.x * 64 + lane;
    const bool nv = n < N;
    const float* qr = p.qk + ((size_t)b * N + min(n, N - 1)) * 2 * D + h * 2 * hd;
    constexpr int MM = 16;  // context tokens per wave (M <= 64)
    float a1[MM], a2[MM];
#pragma unroll
    for (int mm = 0; mm < MM; ++mm) a1[mm] = a2[mm] = 0.f;
    for (int d = 0; d < hd; d += 4) {
        const f32x4 q1 = *reinterpret_cast<const f32x4*>(qr + d), q2 = *reinterpret_cast<const f32x4*>(qr + hd + d);
#pragma unroll
        for (int mm = 0; mm < MM; ++mm) {
            const int m = wave + 4 * mm;
            if (m < M) {
                const f32x4 ka = *reinterpret_cast<const f32x4*>(k1 + m * hd + d), kb = *reinterpret_cast<const f32x4*>(k2 + m * hd + d);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a1[mm] = fmaf(q1[e], ka[e], a1[mm]);
                    a2[mm] = fmaf(q2[e], kb[e], a2[mm]);
                }
            }
        }
    }
#pragma unroll
    for (int mm = 0; mm < MM; ++mm) {
        const int m = wave + 4 * mm;
        if (m < M) {
            s1[lane * (M + 1) + m] = a1[mm] * p.scale;
            if (nv) p.scores_t[((size_t)bh * M + m) * N + n] = a2[mm] * p.scale;
        }
    }
    __syncthreads();
    const float* sr = s1 + lane * (M + 1);
    float mx = -INFINITY;
    for (int m = 0; m < M; ++m) mx = fmaxf(mx, sr[m]);
    float l = 0.f;
    for (int m = 0; m < M; ++m) l += expf(sr[m] - mx);
    const float inv = 1.0f / l;
    for (int g = wave; g < hd / 32; g += 4) {
        float acc[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) acc[c] = 0.f;
        for (int m = 0; m < M; ++m) {
            const float pm = expf(sr[m] - mx) * inv;
            const float* vr = vsrc + m * hd + g * 32;
#pragma unroll
            for (int c = 0; c < 32; c += 4) {
                const f32x4 vv = *reinterpret_cast<const f32x4*>(vr + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[c + e] = fmaf(pm, vv[e], acc[c + e]);
            }
        }
        if (nv) {
            bf16* dst = p.y + a_pos<PLANES>((int64_t)b * N + n, D, h * hd + g * 32);  // one whole 32-column operand block
#pragma unroll
            for (int c = 0; c < 32; c += 8) {
                bf16x8 hv, lv;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bf16 hi = (bf16)acc[c + e];
                    hv[e] = hi;
                    lv[e] = (bf16)(acc[c + e] - (float)hi);
                }
                *reinterpret_cast<bf16x8*>(dst + c) = hv;
                if constexpr (PLANES == 2) *reinterpret_cast<bf16x8*>(dst + kLoOffset + c) = lv;
            }
        }
    }
}

// partial[((bh * kCrossSplit + split) * M + m) * (hd + 2) + {d | hd: local max | hd + 1: local sum}]
template <int MMAX>
__global__ __launch_bounds__(256) void cross_attn_src_partial_kernel(const CrossAttnParams p) {
    __shared__ __attribute__((aligned(16))) float pb[MMAX * 64];
    __shared__ float mxs[MMAX], ls[MMAX];
    const int hd = p.head_dim, D = p.heads * hd, N = p.N, M = p.M;
    const int split = blockIdx.x, bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n_lo = (int)((int64_t)split * N / kCrossSplit), n_hi = (int)((int64_t)(split + 1) * N / kCrossSplit);
    const float* st = p.scores_t + (size_t)bh * M * N;
    for (int m = wave; m < M; m += 4) {  // local max per context token
        float mx = -INFINITY;
        for (int n = n_lo + lane; n < n_hi; n += 64) mx = fmaxf(mx, st[(size_t)m * N + n]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        if (lane == 0) mxs[m] = mx;
    }
    __syncthreads();
    float acc[MMAX], lacc[MMAX / 4];
#pragma unroll
    for (int m = 0; m < MMAX; ++m) acc[m] = 0.f;
#pragma unroll
    for (int i = 0; i < MMAX / 4; ++i) lacc[i] = 0.f;
    const float* vb = p.v + (size_t)b * N * D + h * hd + t;
    for (int n0 = n_lo; n0 < n_hi; n0 += 64) {
#pragma unroll
        for (int i = 0; i < MMAX / 4; ++i) {  // thread (wave, lane) fills p[m = wave + 4 i][j = lane]
            const int m = wave + 4 * i;
            if (m < M) {
                const int n = n0 + lane;
                const float pv = n < n_hi ? expf(st[(size_t)m * N + n] - mxs[m]) : 0.f;
                pb[m * 64 + lane] = pv;
                lacc[i] += pv;
            }
        }
        __syncthreads();
        if (t < hd) {
            for (int j = 0; j < 64; j += 4) {
                float vv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) vv[e] = (n0 + j + e < n_hi) ? vb[(size_t)(n0 + j + e) * D] : 0.f;
#pragma unroll
                for (int m = 0; m < MMAX; ++m) {
                    if (m < M) {
                        const f32x4 pv = *reinterpret_cast<const f32x4*>(pb + m * 64 + j);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[m] = fmaf(pv[e], vv[e], acc[m]);
                    }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MMAX / 4; ++i) {
        const float lsum = wave_sum(lacc[i]);
        if (lane == 0 && wave + 4 * i < M) ls[wave + 4 * i] = lsum;
    }
    __syncthreads();
    float* out = p.partial + ((size_t)(bh * kCrossSplit + split) * M) * (hd + 2);
#pragma unroll
    for (int m = 0; m < MMAX; ++m) {
        if (m < M) {
            if (t < hd) out[(size_t)m * (hd + 2) + t] = acc[m];
            if (t == 0) {
                out[(size_t)m * (hd + 2) + hd] = mxs[m];
                out[(size_t)m * (hd + 2) + hd + 1] = ls[m];
            }
        }
    }
}

template <int PLANES>
__global__ __launch_bounds__(256) void cross_attn_src_combine_kernel(const CrossAttnParams p) {
    const int hd = p.head_dim, D = p.heads * hd, M = p.M;
    const int m = blockIdx.x, bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads, t = threadIdx.x;
    if (t >= hd) return;
    const float* part = p.partial + ((size_t)bh * kCrossSplit * M + m) * (hd + 2);
    const size_t sstride = (size_t)M * (hd + 2);
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < kCrossSplit; ++s) mx = fmaxf(mx, part[s * sstride + hd]);
    float l = 0.f, acc = 0.f;
#pragma unroll
    for (int s = 0; s < kCrossSplit; ++s) {
        const float w = expf(part[s * sstride + hd] - mx);  // an empty split has max = -inf: weight 0
        l = fmaf(part[s * sstride + hd + 1], w, l);
        acc = fmaf(part[s * sstride + t], w, acc);
    }
    bf16 hi, lo;
    split_bf16(acc / l, hi, lo);
    bf16* dst = p.y_src + a_pos<PLANES>((int64_t)b * M + m, D, h * hd + t);
    *dst = hi;
    if constexpr (PLANES == 2) dst[kLoOffset] = lo;
}

size_t cross_attention_partial_floats(int B, int heads, int M, int head_dim) { return (size_t)B * heads * kCrossSplit * M * (head_dim + 2); }

int launch_cross_attention(const CrossAttnParams& p, int planes, hipStream_t stream) {
    CWM_REQUIRE(p.M > 0 && p.M <= 64 && p.head_dim > 0 && p.head_dim <= 256 && p.head_dim % 32 == 0,
                "cross_attention: needs M <= 64 and head_dim a multiple of 32, <= 256 (got %d, %d)", p.M, p.head_dim);
    CWM_REQUIRE(p.scores_t && p.partial, "cross_attention: scratch buffers missing");
    const size_t smem = ((size_t)3 * p.M * p.head_dim + (size_t)64 * (p.M + 1)) * sizeof(float);
    const dim3 g1((p.N + 63) / 64, p.B * p.heads), g2(kCrossSplit, p.B * p.heads), g3(p.M, p.B * p.heads);
    if (planes == 1) {
        if (int rc = cwm_set_max_lds((const void*)cross_attn_main_kernel<1>, 160 * 1024)) return rc;  // per (device, kernel): engine.hip
        hipLaunchKernelGGL(cross_attn_main_kernel<1>, g1, dim3(256), smem, stream, p);
    } else {
        if (int rc = cwm_set_max_lds((const void*)cross_attn_main_kernel<2>, 160 * 1024)) return rc;
        hipLaunchKernelGGL(cross_attn_main_kernel<2>, g1, dim3(256), smem, stream, p);
    }
    if (p.M <= 32) hipLaunchKernelGGL(cross_attn_src_partial_kernel<32>, g2, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(cross_attn_src_partial_kernel<64>, g2, dim3(256), 0, stream, p);
    if (planes == 1) hipLaunchKernelGGL(cross_attn_src_combine_kernel<1>, g3, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(cross_attn_src_combine_kernel<2>, g3, dim3(256), 0, stream, p);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace cwm
