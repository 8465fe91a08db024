// Kernels specific to the IMU-conditioned conjoined padded predictor (BASELINE configs[4]):
// null-token padding bookkeeping, IMU tubelet gather, short-sequence self-attention for the context
// stream (head_dim 32, <= 64 tokens) and the bidirectional cross attention between the N-token RGB
// stream and the M <= 64-token IMU stream.  The cross/small attentions are exact-fp32 VALU kernels:
// together they are < 5 % of this model's time (the RGB stream's 6336-token self-attention dominates).
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
// Kernel 1: one wave per main token (4 tokens / workgroup), context K1/K2/V_src of the (b,h) pair in LDS;
//           lanes split the head dimension, scores via wave reductions; writes y and scores_t[b,h,m,n].
// Kernel 2: one workgroup per (b,h,m): softmax over n of scores_t, then the weighted sum of v rows.
// ---------------------------------------------------------------------------------------------
template <int PLANES>
__global__ __launch_bounds__(256) void cross_attn_main_kernel(const CrossAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int hd = p.head_dim, M = p.M, D = p.heads * hd;
    float* k1 = lds;                 // [M][hd]
    float* k2 = k1 + (size_t)M * hd; // [M][hd]
    float* vsrc = k2 + (size_t)M * hd;
    const int bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads;
    for (int i = threadIdx.x; i < M * hd; i += 256) {
        const int m = i / hd, d = i - m * hd;
        const float* r = p.qk_src + ((size_t)b * M + m) * 2 * D + h * 2 * hd;
        k1[i] = r[d];
        k2[i] = r[hd + d];
        vsrc[i] = p.v_src[((size_t)b * M + m) * D + h * hd + d];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    if (n >= p.N) return;
    constexpr int PER = 3;  // head_dim <= 192
    float q1[PER], q2[PER];
    const float* qr = p.qk + ((size_t)b * p.N + n) * 2 * D + h * 2 * hd;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int d = lane + 64 * i;
        q1[i] = d < hd ? qr[d] * p.scale : 0.f;
        q2[i] = d < hd ? qr[hd + d] * p.scale : 0.f;
    }
    float my = -INFINITY;  // lane m keeps score m (M <= 64): no runtime-indexed register array
    float* st = p.scores_t + ((size_t)bh * M) * p.N + n;
#pragma unroll 1
    for (int m = 0; m < M; ++m) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int d = lane + 64 * i;
            if (d < hd) {
                a1 = fmaf(q1[i], k1[m * hd + d], a1);
                a2 = fmaf(q2[i], k2[m * hd + d], a2);
            }
        }
        a1 = wave_sum(a1);
        a2 = wave_sum(a2);
        if (lane == m) my = a1;
        if (lane == 0) st[(size_t)m * p.N] = a2;
    }
    float mx = my;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const float pl = lane < M ? expf(my - mx) : 0.f;
    const float l = wave_sum(pl);
    float o[PER] = {0.f, 0.f, 0.f};
#pragma unroll 1
    for (int m = 0; m < M; ++m) {
        const float pm = __shfl(pl, m, 64);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int d = lane + 64 * i;
            if (d < hd) o[i] = fmaf(pm, vsrc[m * hd + d], o[i]);
        }
    }
    const float inv = 1.0f / l;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int d = lane + 64 * i;
        if (d < hd) {
            bf16 hi, lo;
            split_bf16(o[i] * inv, hi, lo);
            bf16* dst = p.y + a_pos<PLANES>((int64_t)b * p.N + n, D, h * hd + d);
            *dst = hi;
            if constexpr (PLANES == 2) dst[kLoOffset] = lo;
        }
    }
}

template <int PLANES>
__global__ __launch_bounds__(256) void cross_attn_src_kernel(const CrossAttnParams p) {
    __shared__ float red[256];
    __shared__ float pbuf[256];
    const int hd = p.head_dim, D = p.heads * hd, N = p.N;
    const int m = blockIdx.x, bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads;
    const int t = threadIdx.x;
    const float* st = p.scores_t + ((size_t)bh * p.M + m) * N;
    float mx = -INFINITY;
    for (int n = t; n < N; n += 256) mx = fmaxf(mx, st[n]);
    red[t] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) red[t] = fmaxf(red[t], red[t + s]);
        __syncthreads();
    }
    mx = red[0];
    __syncthreads();
    float l = 0.f;
    float acc = 0.f;  // thread t < hd accumulates output feature t
    for (int n0 = 0; n0 < N; n0 += 256) {
        const int n = n0 + t;
        const float pm = n < N ? expf(st[n] - mx) : 0.f;
        pbuf[t] = pm;
        l += pm;
        __syncthreads();
        if (t < hd) {
            const int cnt = min(256, N - n0);
            const float* vb = p.v + ((size_t)b * N + n0) * D + h * hd + t;
            for (int j = 0; j < cnt; ++j) acc = fmaf(pbuf[j], vb[(size_t)j * D], acc);
        }
        __syncthreads();
    }
    red[t] = l;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    if (t < hd) {
        bf16 hi, lo;
        split_bf16(acc / red[0], hi, lo);
        bf16* dst = p.y_src + a_pos<PLANES>((int64_t)b * p.M + m, D, h * hd + t);
        *dst = hi;
        if constexpr (PLANES == 2) dst[kLoOffset] = lo;
    }
}

int launch_cross_attention(const CrossAttnParams& p, int planes, hipStream_t stream) {
    CWM_REQUIRE(p.M > 0 && p.M <= 64 && p.head_dim > 0 && p.head_dim <= 192, "cross_attention: needs M <= 64 and head_dim <= 192 (got %d, %d)", p.M, p.head_dim);
    const size_t smem = (size_t)3 * p.M * p.head_dim * sizeof(float);
    const dim3 g1((p.N + 3) / 4, p.B * p.heads), g2(p.M, p.B * p.heads);
    if (planes == 1) {
        static bool a1 = false;
        if (!a1) { CWM_HIP_CHECK(hipFuncSetAttribute((const void*)cross_attn_main_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); a1 = true; }
        hipLaunchKernelGGL(cross_attn_main_kernel<1>, g1, dim3(256), smem, stream, p);
        hipLaunchKernelGGL(cross_attn_src_kernel<1>, g2, dim3(256), 0, stream, p);
    } else {
        static bool a2 = false;
        if (!a2) { CWM_HIP_CHECK(hipFuncSetAttribute((const void*)cross_attn_main_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); a2 = true; }
        hipLaunchKernelGGL(cross_attn_main_kernel<2>, g1, dim3(256), smem, stream, p);
        hipLaunchKernelGGL(cross_attn_src_kernel<2>, g2, dim3(256), 0, stream, p);
    }
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace cwm
