// Flow-sample statistics on the device: the reductions the reference runs over the S counterfactual flow
// samples right after the predictor path.
//
// Replaces (cwm/models/segmentation.py):
//   FlowGenerator.compute_flow_corrs             :479-547  ds x ds average pool, ChannelMSE against zeros (utils.py:510-513), the optional
//                                                          prologues (:519-538: Spearman argsort, thresholds, normalise, z-score),
//                                                          torch.cov / torch.corrcoef over the samples, NaN -> 0
//   FlowGenerator.compute_flow_samples_magnitude :250-255
//   FlowGenerator.compute_mean_motion_map        :257-276
// All arithmetic is fp32 (the reference's dtype).  The covariance is a [P, S] x [S, P] product with P = (H/ds)(W/ds) up to
// 12544 and S = 8 .. 256: 629 MB of output per frame pair at ds = 2, i.e. bound by the HBM write below S ~ 64 and by the
// fp32 matrix pipe above; it runs on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate) with the operands swapped
// so that a lane owns 4 consecutive columns of one output row (16-byte stores).  Sharding: rows [row0, row0 + nrows) per
// call, so that ranks which all-gather the small feature matrix each produce a row slab (dist.py).
#include "../../include/cwm_hip.h"
#include "common.h"
#include "kernels.h"

namespace cwm {

// ---- features: X[b][p][s] = sqrt(mean_c(avgpool_ds(flow[b][c])^2)) ----------------------------------------------------
__global__ void flow_features_kernel(const float* __restrict__ f, int64_t sb, int64_t sc, int64_t sh, int64_t sw, int64_t ss, int B, int C, int H,
                                     int W, int S, int ds, float* __restrict__ x) {
    const int Wd = W / ds, P = (H / ds) * Wd;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * P * S) return;
    const int s = (int)(i % S);
    const int64_t bp = i / S;
    const int pidx = (int)(bp % P), b = (int)(bp / P);
    const int py = pidx / Wd, px = pidx - py * Wd;
    const float inv_area = 1.0f / (float)(ds * ds);
    float acc = 0.f;
    for (int c = 0; c < C; ++c) {
        const float* fc = f + b * sb + c * sc + s * ss;
        float pool = 0.f;
        for (int dy = 0; dy < ds; ++dy)
            for (int dx = 0; dx < ds; ++dx) pool += fc[(int64_t)(py * ds + dy) * sh + (int64_t)(px * ds + dx) * sw];
        pool *= inv_area;
        acc += pool * pool;
    }
    x[i] = sqrtf(acc / (float)C);
}

// ---- centre the rows; inverse standard deviations for the correlation form ----------------------------------------------
// one wave per row (b, p): xc = x - mean_S(x); inv_std = 1 / sqrt(sum xc^2 / (S - 1))  (the scale of torch.corrcoef)
__global__ void flow_center_kernel(const float* __restrict__ x, int rows, int S, float* __restrict__ xc, float* __restrict__ inv_std) {
    const int row = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * S;
    float sum = 0.f;
    for (int s = lane; s < S; s += 64) sum += xr[s];
    const float mean = wave_sum(sum) / (float)S;
    float sq = 0.f;
    for (int s = lane; s < S; s += 64) {
        const float d = xr[s] - mean;
        xc[(size_t)row * S + s] = d;
        sq += d * d;
    }
    sq = wave_sum(sq);
    if (lane == 0) inv_std[row] = 1.0f / sqrtf(sq / (float)(S - 1));
}

// ---- covariance / correlation slab: out[b][i - row0][j] = scale * sum_s xc[b][i][s] xc[b][j][s] ---------------------------
// 128 x 128 output tile per 256-thread workgroup, 64 x 64 per wave = 4 x 4 fragments of v_mfma_f32_16x16x4_f32.
// D^T orientation: A operand = the j rows, B operand = the i rows, so lane l holds out[i = 16 fi + l % 16][j = 16 fj + 4 (l / 16) + r].
constexpr int kCovBK = 32;  // samples staged per step
__global__ __launch_bounds__(256) void flow_cov_kernel(const float* __restrict__ xc, const float* __restrict__ inv_std, int P, int S, int row0,
                                                       int nrows, int use_cov, float* __restrict__ out) {
    __shared__ float lds_i[128][kCovBK + 1];
    __shared__ float lds_j[128][kCovBK + 1];
    const int b = blockIdx.z;
    const int i0 = row0 + blockIdx.y * 128, j0 = blockIdx.x * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const float* xb = xc + (size_t)b * P * S;

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < S; k0 += kCovBK) {
        // stage 128 rows x 32 samples of both operands (zero fill past P / S)
        for (int e = tid; e < 128 * kCovBK; e += 256) {
            const int r = e / kCovBK, k = e - r * kCovBK;
            const bool kin = k0 + k < S;
            lds_i[r][k] = (kin && i0 + r < P) ? xb[(size_t)(i0 + r) * S + k0 + k] : 0.f;
            lds_j[r][k] = (kin && j0 + r < P) ? xb[(size_t)(j0 + r) * S + k0 + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kCovBK; kk += 4) {
            const int kq = kk + (lane >> 4);
            float av[4], bv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                av[t] = lds_j[wj * 64 + t * 16 + (lane & 15)][kq];  // A operand: rows = j
                bv[t] = lds_i[wi * 64 + t * 16 + (lane & 15)][kq];  // B operand: columns = i
            }
#pragma unroll
            for (int fi = 0; fi < 4; ++fi)
#pragma unroll
                for (int fj = 0; fj < 4; ++fj) acc[fi][fj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[fj], bv[fi], acc[fi][fj], 0, 0, 0);
        }
        __syncthreads();
    }

    const float inv_n1 = 1.0f / (float)(S - 1);
#pragma unroll
    for (int fi = 0; fi < 4; ++fi) {
        const int i = i0 + wi * 64 + fi * 16 + (lane & 15);
        if (i >= P || i >= row0 + nrows) continue;
        const float si = use_cov ? 1.0f : inv_std[(size_t)b * P + i];
#pragma unroll
        for (int fj = 0; fj < 4; ++fj) {
            const int j = j0 + wj * 64 + fj * 16 + (lane >> 4) * 4;
            if (j >= P) continue;
            f32x4 v = acc[fi][fj] * inv_n1;  // torch.cov: unbiased (correction = 1)
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float c = v[r];
                if (!use_cov && j + r < P) {
                    c = c * si * inv_std[(size_t)b * P + j + r];
                    // torch.corrcoef clips to [-1, 1] and keeps a NaN (a constant row: 0 * inf) a NaN; fminf / fmaxf would turn it into -1
                    if (c == c) c = fminf(1.0f, fmaxf(-1.0f, c));
                }
                o[r] = (c != c) ? 0.f : c;  // NaN -> 0 (segmentation.py:541)
            }
            float* dst = out + ((size_t)b * nrows + (i - row0)) * P + j;
            if (j + 3 < P && (P & 3) == 0) {
                *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (j + r < P) dst[r] = o[r];
            }
        }
    }
}

// ---- feature prologues of compute_flow_corrs (segmentation.py:519-538), in place on x[b][p][s] ------------------------------------
// The reference applies them to flow_inp_b = x[b] as a [P, S] matrix: the Spearman form replaces every ROW (position) by
// argsort over its samples; the threshold / range / normalise / z-score forms use statistics over dim 0 -- over the P POSITIONS,
// separately for every sample column s.

// argsort of every row over its S samples, as floats (`torch.argsort(flow_inp[b], -1).float()`, :521): out[rank of element j] = j.
// One wave per row; rank = #(smaller) + #(equal with a lower index): the stable order (torch's default sort is not stable, so rows
// with exactly tied values are the one place where the reference's own output is implementation-defined).  NaNs order LAST and equal to each
// other, as torch.sort places them: the result is a permutation of 0 .. S-1 for every input.
__global__ __launch_bounds__(256) void flow_argsort_rows_kernel(float* __restrict__ x, int rows, int S) {
    extern __shared__ float srow[];  // [4 waves][S]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    float* mine = srow + (size_t)wave * S;
    if (row < rows)
        for (int s = lane; s < S; s += 64) mine[s] = x[(size_t)row * S + s];
    __syncthreads();
    if (row >= rows) return;
    for (int j = lane; j < S; j += 64) {
        const float v = mine[j];
        int rank = 0;
        for (int k = 0; k < S; ++k) {
            const float u = mine[k];
            const bool less = u < v || (v != v && u == u);
            const bool same = u == v || (u != u && v != v);
            rank += (less || (same && k < j)) ? 1 : 0;
        }
        x[(size_t)row * S + rank] = (float)j;
    }
}

// per column (b, s): {min, max, mean, unbiased std} over the P positions -> st[b][s]; sums in float64 (mean first, then the
// squared deviations: two passes over a column that stays in L2)
__global__ __launch_bounds__(256) void flow_colstats_kernel(const float* __restrict__ x, int P, int S, float4* __restrict__ st) {
    __shared__ float rmn[4][64], rmx[4][64];
    __shared__ double rs[4][64];
    const int b = blockIdx.y, s = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const bool in = s < S;
    const float* xb = x + (size_t)b * P * S + (in ? s : 0);
    float mn = INFINITY, mx = -INFINITY;
    double sum = 0.0;  // (a NaN in the column makes the sum, and with it min / max / mean / std, NaN: torch.amin / amax / mean / std propagate it)
    if (in)
        for (int p = rg; p < P; p += 4) {
            const float v = xb[(size_t)p * S];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
            sum += (double)v;
        }
    rmn[rg][threadIdx.x & 63] = mn;
    rmx[rg][threadIdx.x & 63] = mx;
    rs[rg][threadIdx.x & 63] = sum;
    __syncthreads();
    const int c = threadIdx.x & 63;
    const double mean = (rs[0][c] + rs[1][c] + rs[2][c] + rs[3][c]) / (double)P;
    __syncthreads();
    double sq = 0.0;
    if (in)
        for (int p = rg; p < P; p += 4) {
            const double d = (double)xb[(size_t)p * S] - mean;
            sq += d * d;
        }
    rs[rg][c] = sq;
    __syncthreads();
    if (rg == 0 && in) {
        const double var = (rs[0][c] + rs[1][c] + rs[2][c] + rs[3][c]) / (double)(P - 1);  // P == 1: 0 / 0 = NaN, as torch.std
        const bool has_nan = mean != mean;
        const float qn = __builtin_nanf("");
        st[(size_t)b * S + s] = make_float4(has_nan ? qn : fminf(fminf(rmn[0][c], rmn[1][c]), fminf(rmn[2][c], rmn[3][c])),
                                            has_nan ? qn : fmaxf(fmaxf(rmx[0][c], rmx[1][c]), fmaxf(rmx[2][c], rmx[3][c])), (float)mean, (float)sqrt(var));
    }
}

// elementwise forms; op: 1 x * (x > a), 2 (x > a), 3 ((x - min) > a * (max - min)), 4 x / max(colmax, eps), 5 (x - mean) / max(std, eps)
__global__ void flow_apply_kernel(float* __restrict__ x, int64_t total, int P, int S, int op, float a, float eps, const float4* __restrict__ st) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float v = x[i];
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (op >= 3) c = st[(i / ((int64_t)P * S)) * S + (i % S)];
    float o;
    if (op == 1) o = v * ((v > a) ? 1.0f : 0.0f);
    else if (op == 2) o = (v > a) ? 1.0f : 0.0f;
    else if (op == 3) o = ((v - c.x) > a * (c.y - c.x)) ? 1.0f : 0.0f;
    else if (op == 4) o = v / (c.y != c.y ? c.y : fmaxf(c.y, eps));   // tensor.clamp(min=eps) keeps a NaN
    else o = (v - c.z) / (c.w != c.w ? c.w : fmaxf(c.w, eps));
    x[i] = o;
}

// ---- motion maps ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float flow_mag(const float* f, int64_t sc, int C) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) {
        const float v = f[c * sc];
        a += v * v;
    }
    return sqrtf(a);
}

__device__ __forceinline__ void block_minmax(float& mn, float& mx, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off, 64));
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    }
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[wave] = mn;
        red[16 + wave] = mx;
    }
    __syncthreads();
    mn = red[0];
    mx = red[16];
    for (int w = 1; w < nw; ++w) {
        mn = fminf(mn, red[w]);
        mx = fmaxf(mx, red[16 + w]);
    }
}

// per (b, s): min and max over (H, W) of the flow magnitude -> mm[b][s] = {min, max}
__global__ void flow_mag_minmax_kernel(const float* __restrict__ f, int64_t sb, int64_t sc, int64_t sh, int64_t sw, int64_t ss, int C, int H, int W,
                                       int S, float2* __restrict__ mm) {
    __shared__ float red[32];
    const int s = blockIdx.x, b = blockIdx.y;
    const float* fb = f + b * sb + s * ss;
    float mn = INFINITY, mx = -INFINITY;
    for (int e = threadIdx.x; e < H * W; e += blockDim.x) {
        const int y = e / W, x = e - y * W;
        const float m = flow_mag(fb + (int64_t)y * sh + (int64_t)x * sw, sc, C);
        mn = fminf(mn, m);
        mx = fmaxf(mx, m);
    }
    block_minmax(mn, mx, red);
    if (threadIdx.x == 0) mm[(size_t)b * S + s] = make_float2(mn, mx);
}

// sum over the S samples of the (optionally per-sample range-normalised) magnitude -> sum[b][y][x]
__global__ void flow_motion_sum_kernel(const float* __restrict__ f, int64_t sb, int64_t sc, int64_t sh, int64_t sw, int64_t ss, int B, int C, int H,
                                       int W, int S, const float2* __restrict__ mm, float eps, float* __restrict__ sum) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * H * W) return;
    const int b = (int)(i / (H * W)), e = (int)(i - (int64_t)b * H * W);
    const int y = e / W, x = e - y * W;
    const float* fp = f + b * sb + (int64_t)y * sh + (int64_t)x * sw;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) {
        float m = flow_mag(fp + s * ss, sc, C);
        if (mm) {
            const float2 r = mm[(size_t)b * S + s];
            m = (m - r.x) / fmaxf(r.y - r.x, eps);
        }
        acc += m;
    }
    sum[i] = acc;
}

// map = sum * scale; if normalize: (map - min) / max(max - min, eps) over (H, W) per b.  One workgroup per b.
__global__ void flow_map_finish_kernel(float* __restrict__ map, int HW, float scale, int normalize, float eps) {
    __shared__ float red[32];
    float* m = map + (size_t)blockIdx.x * HW;
    float mn = INFINITY, mx = -INFINITY;
    for (int e = threadIdx.x; e < HW; e += blockDim.x) {
        const float v = m[e] * scale;
        m[e] = v;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    if (!normalize) return;
    block_minmax(mn, mx, red);
    const float d = fmaxf(mx - mn, eps);
    for (int e = threadIdx.x; e < HW; e += blockDim.x) m[e] = (m[e] - mn) / d;
}

}  // namespace cwm

using namespace cwm;

extern "C" int cwm_flow_features(const float* flows_dev, const int64_t* strides, int B, int C, int H, int W, int S, int downsample, float* x_dev,
                                 void* stream) {
    CWM_REQUIRE(flows_dev && strides && x_dev && B > 0 && C > 0 && H > 0 && W > 0 && S > 0, "cwm_flow_features: bad argument");
    CWM_REQUIRE(downsample >= 1 && H % downsample == 0 && W % downsample == 0, "cwm_flow_features: downsample=%d must divide H=%d and W=%d",
                downsample, H, W);
    const int64_t total = (int64_t)B * (H / downsample) * (W / downsample) * S;
    hipLaunchKernelGGL(flow_features_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flows_dev, strides[0],
                       strides[1], strides[2], strides[3], strides[4], B, C, H, W, S, downsample, x_dev);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int cwm_flow_cov(const float* x_dev, int B, int P, int S, int row0, int nrows, int use_covariance, float* xc_work_dev,
                            float* inv_std_work_dev, float* out_dev, void* stream) {
    CWM_REQUIRE(x_dev && xc_work_dev && inv_std_work_dev && out_dev && B > 0 && P > 0 && S > 0, "cwm_flow_cov: bad argument");
    CWM_REQUIRE(row0 >= 0 && nrows > 0 && row0 + nrows <= P, "cwm_flow_cov: rows [%d, %d) outside [0, %d)", row0, row0 + nrows, P);
    hipStream_t s = (hipStream_t)stream;
    const int rows = B * P;
    hipLaunchKernelGGL(flow_center_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x_dev, rows, S, xc_work_dev, inv_std_work_dev);
    const dim3 grid((unsigned)((P + 127) / 128), (unsigned)((nrows + 127) / 128), (unsigned)B);
    hipLaunchKernelGGL(flow_cov_kernel, grid, dim3(256), 0, s, xc_work_dev, inv_std_work_dev, P, S, row0, nrows, use_covariance ? 1 : 0, out_dev);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int cwm_flow_transform(float* x_dev, int B, int P, int S, int spearman, int thresh_mode, float thresh, int normalize, int zscore, float eps,
                                  float* stats_work_dev, void* stream) {
    CWM_REQUIRE(x_dev && B > 0 && P > 0 && S > 0, "cwm_flow_transform: bad argument");
    CWM_REQUIRE(thresh_mode >= 0 && thresh_mode <= 3, "cwm_flow_transform: thresh_mode must be 0 (none), 1 (x * (x > t)), 2 (x > t) or 3 (range threshold)");
    CWM_REQUIRE(!(thresh_mode == 3 || normalize || zscore) || stats_work_dev, "cwm_flow_transform: the column statistics need the [B][S][4] work buffer");
    CWM_REQUIRE(!spearman || S <= 4096, "cwm_flow_transform: Spearman ranks support at most 4096 samples (S=%d)", S);
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * P * S;
    const unsigned eblocks = (unsigned)((total + 255) / 256);
    float4* st = reinterpret_cast<float4*>(stats_work_dev);
    const dim3 sgrid((unsigned)((S + 63) / 64), (unsigned)B);
    if (spearman) {
        const size_t smem = (size_t)4 * S * sizeof(float);
        if (smem > 48 * 1024)
            if (int rc = cwm_set_max_lds((const void*)flow_argsort_rows_kernel, (int)smem)) return rc;
        hipLaunchKernelGGL(flow_argsort_rows_kernel, dim3((unsigned)((B * P + 3) / 4)), dim3(256), smem, s, x_dev, B * P, S);
    }
    if (thresh_mode == 1 || thresh_mode == 2) {
        hipLaunchKernelGGL(flow_apply_kernel, dim3(eblocks), dim3(256), 0, s, x_dev, total, P, S, thresh_mode, thresh, eps, st);
    } else if (thresh_mode == 3) {
        hipLaunchKernelGGL(flow_colstats_kernel, sgrid, dim3(256), 0, s, x_dev, P, S, st);
        hipLaunchKernelGGL(flow_apply_kernel, dim3(eblocks), dim3(256), 0, s, x_dev, total, P, S, 3, thresh, eps, st);
    }
    if (normalize) {
        hipLaunchKernelGGL(flow_colstats_kernel, sgrid, dim3(256), 0, s, x_dev, P, S, st);
        hipLaunchKernelGGL(flow_apply_kernel, dim3(eblocks), dim3(256), 0, s, x_dev, total, P, S, 4, 0.f, eps, st);
    }
    if (zscore) {
        hipLaunchKernelGGL(flow_colstats_kernel, sgrid, dim3(256), 0, s, x_dev, P, S, st);
        hipLaunchKernelGGL(flow_apply_kernel, dim3(eblocks), dim3(256), 0, s, x_dev, total, P, S, 5, 0.f, eps, st);
    }
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int cwm_flow_motion_sum(const float* flows_dev, const int64_t* strides, int B, int C, int H, int W, int S, int normalize_per_sample,
                                   float eps, float* minmax_work_dev, float* sum_dev, void* stream) {
    CWM_REQUIRE(flows_dev && strides && sum_dev && B > 0 && C > 0 && H > 0 && W > 0 && S > 0, "cwm_flow_motion_sum: bad argument");
    CWM_REQUIRE(!normalize_per_sample || minmax_work_dev, "cwm_flow_motion_sum: per-sample normalisation needs the [B][S][2] work buffer");
    hipStream_t s = (hipStream_t)stream;
    float2* mm = nullptr;
    if (normalize_per_sample) {
        mm = reinterpret_cast<float2*>(minmax_work_dev);
        hipLaunchKernelGGL(flow_mag_minmax_kernel, dim3((unsigned)S, (unsigned)B), dim3(256), 0, s, flows_dev, strides[0], strides[1], strides[2],
                           strides[3], strides[4], C, H, W, S, mm);
    }
    const int64_t total = (int64_t)B * H * W;
    hipLaunchKernelGGL(flow_motion_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, flows_dev, strides[0], strides[1],
                       strides[2], strides[3], strides[4], B, C, H, W, S, mm, eps, sum_dev);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int cwm_flow_map_finish(float* map_dev, int B, int HW, float scale, int normalize, float eps, void* stream) {
    CWM_REQUIRE(map_dev && B > 0 && HW > 0, "cwm_flow_map_finish: bad argument");
    hipLaunchKernelGGL(flow_map_finish_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, map_dev, HW, scale, normalize, eps);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}
