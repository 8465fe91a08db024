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

// the same with the sample axis innermost (ss == 1) and S a multiple of 4: four samples per thread, 16-byte loads (a quarter of the load instructions)
__global__ void flow_features4_kernel(const float* __restrict__ f, int64_t sb, int64_t sc, int64_t sh, int64_t sw, int B, int C, int H, int W, int S, int ds,
                                      float* __restrict__ x) {
    const int Wd = W / ds, P = (H / ds) * Wd, S4 = S / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * P * S4) return;
    const int s = (int)(i % S4) * 4;
    const int64_t bp = i / S4;
    const int pidx = (int)(bp % P), b = (int)(bp / P);
    const int py = pidx / Wd, px = pidx - py * Wd;
    const float inv_area = 1.0f / (float)(ds * ds);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
        const float* fc = f + b * sb + c * sc + s;
        f32x4 pool = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int dy = 0; dy < ds; ++dy)
            for (int dx = 0; dx < ds; ++dx) pool += *reinterpret_cast<const f32x4*>(fc + (int64_t)(py * ds + dy) * sh + (int64_t)(px * ds + dx) * sw);
        pool = pool * inv_area;
        acc += pool * pool;
    }
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = sqrtf(acc[r] / (float)C);
    *reinterpret_cast<f32x4*>(x + ((size_t)b * P + pidx) * S + s) = o;
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
// 128 x 128 output tile per 256-thread workgroup, 64 x 64 per wave = 4 x 4 fragments of v_mfma_f32_16x16x4_f32 (exact fp32 products).
// D^T orientation: A operand = the j rows, B operand = the i rows, so lane l holds out[i = 16 fi + l % 16][j = 16 fj + 4 (l / 16) + r].
// Round 6 (profiles/r6_bench_flowstats*.json: 1.45 ms at S = 256 = 0.35 of the fp32 matrix peak, 281 us at S = 24 = 2.2 TB/s of writes):
//  * the matrix is symmetric: a call for the WHOLE matrix (row0 = 0, nrows = P) computes the tiles on and above the diagonal only and writes every
//    off-diagonal tile twice, once transposed -- half the MFMA work, the same bytes (SYM).  Row slabs (the sharded form) keep the rectangular grid;
//  * operands come in as 16-byte loads into registers one K step AHEAD of the MFMAs that consume them (the loads of step k + 1 are in flight under the
//    MFMAs of step k; until round 5 every step began with an exposed global -> LDS round trip between two barriers);
//  * the epilogue goes through LDS so that every store instruction writes whole 256-byte row segments in both orientations (until round 5: 16 rows x 64 B
//    per instruction straight from the accumulators).
constexpr int kCovBK = 32;     // samples staged per step
constexpr int kCovPitch = 36;  // floats per staged operand row: 16-byte aligned rows, and rows r, r + 1 sit 36 mod 64 banks apart (fragment reads conflict-free)
constexpr int kCovEpiPitch = 68;

// FULL: all 128 rows and all 32 samples of the step are in range (a workgroup-uniform condition: every step of every tile at P = 12544, S = 256) -- eight plain
// 16-byte loads, nothing between them.  Otherwise out-of-range elements are zero.  (With the bounds checks around every load hipcc put an `s_waitcnt vmcnt(0)`
// between consecutive loads -- the zeroing writes the load's own destination registers --: eight serialised L2 round trips per K step and the matrix pipe 47 % busy,
// profiles/r6_pmc_flow_cov_before.txt.)
template <bool VEC, bool FULL>
__device__ __forceinline__ unsigned cov_load_step(const float* __restrict__ xb, int P, int S, int r0, int k0, int tid, f32x4 (&v)[4]) {
    unsigned valid = 0xFu;  // bit n: load n is in range (VEC, not FULL: applied when the step is stored to LDS -- see cov_store_step)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int e = tid + 256 * n, r = e >> 3, k = k0 + (e & 7) * 4;
        const int row = r0 + r;
        if constexpr (FULL && VEC) {
            v[n] = *reinterpret_cast<const f32x4*>(xb + (size_t)row * S + k);
        } else if constexpr (VEC) {
            // an UNCONDITIONAL load from a clamped address; whether the element counts is decided later.  (Zeroing the destination right here makes hipcc wait for
            // the load -- `vmcnt(0)` between consecutive loads, 16 serialised L2 round trips per step: at S = 24 that was most of the launch.)
            v[n] = *reinterpret_cast<const f32x4*>(xb + (size_t)min(row, P - 1) * S + min(k, S - 4));  // (S % 4 == 0: the four samples are in range together)
            if (!(row < P && k < S)) valid &= ~(1u << n);
        } else {
            f32x4 q = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < P) {
                const float* src = xb + (size_t)row * S + k;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (k + c < S) q[c] = src[c];
            }
            v[n] = q;
        }
    }
    return valid;
}

// MODE 2: the launch has only whole tiles and whole steps (P % 128 == 0, S % 32 == 0, slabs starting on a tile row: decided on the host) -- the kernel contains
// no bounds path at all (a run-time choice between the two forms inside one kernel merges their registers at the join, and the copies there wait for the loads);
// MODE 1: 16-byte loads with bounds; MODE 0: S is not a multiple of 4.
template <int MODE>
__device__ __forceinline__ unsigned cov_load_pair(const float* __restrict__ xb, int P, int S, int i0, int j0, int k0, int tid, f32x4 (&vi)[4], f32x4 (&vj)[4]) {
    const unsigned a = cov_load_step<(MODE >= 1), (MODE == 2)>(xb, P, S, i0, k0, tid, vi);
    const unsigned b = cov_load_step<(MODE >= 1), (MODE == 2)>(xb, P, S, j0, k0, tid, vj);
    return a | (b << 4);
}

__device__ __forceinline__ void cov_store_step(float* lds, int tid, const f32x4 (&v)[4], unsigned valid) {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int e = tid + 256 * n;
        *reinterpret_cast<f32x4*>(lds + (e >> 3) * kCovPitch + (e & 7) * 4) = (valid >> n) & 1u ? v[n] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

template <bool SYM, int MODE>
__global__ __launch_bounds__(256) void flow_cov_kernel(const float* __restrict__ xc, const float* __restrict__ inv_std, int P, int S, int row0, int nrows,
                                                       int use_cov, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 128 * kCovPitch];  // operand tiles i | j; the epilogue's four wave buffers afterwards
    float* lds_i = lds;
    float* lds_j = lds + 128 * kCovPitch;
    const int b = blockIdx.z;
    int ti, tj;
    if constexpr (SYM) {  // linear id -> (ti, tj), tj >= ti, rows of the upper triangle in order
        const int T = (P + 127) / 128, id = blockIdx.x;
        int t = (int)(((2.0f * T + 1.0f) - sqrtf((2.0f * T + 1.0f) * (2.0f * T + 1.0f) - 8.0f * (float)id)) * 0.5f);
        t = max(0, min(t, T - 1));
        while (t > 0 && t * T - t * (t - 1) / 2 > id) --t;                   // first id of row t: t T - t (t - 1) / 2
        while (t + 1 < T && (t + 1) * T - (t + 1) * t / 2 <= id) ++t;
        ti = t;
        tj = t + (id - (t * T - t * (t - 1) / 2));
    } else {
        ti = blockIdx.y;
        tj = blockIdx.x;
    }
    const int i0 = row0 + ti * 128, j0 = tj * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const float* xb = xc + (size_t)b * P * S;

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 vi[4], vj[4];
    unsigned valid = cov_load_pair<MODE>(xb, P, S, i0, j0, 0, tid, vi, vj);
    cov_store_step(lds_i, tid, vi, valid);
    cov_store_step(lds_j, tid, vj, valid >> 4);
    __syncthreads();
    for (int k0 = 0; k0 < S; k0 += kCovBK) {
        const bool more = k0 + kCovBK < S;
        if (more) {  // next step's operands: in flight under this step's MFMAs
            valid = cov_load_pair<MODE>(xb, P, S, i0, j0, k0 + kCovBK, tid, vi, vj);
        }
#pragma unroll 2
        for (int kk = 0; kk < kCovBK; kk += 4) {
            const int kq = kk + (lane >> 4);
            float av[4], bv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                av[t] = lds_j[(wj * 64 + t * 16 + (lane & 15)) * kCovPitch + kq];  // A operand: rows = j
                bv[t] = lds_i[(wi * 64 + t * 16 + (lane & 15)) * kCovPitch + kq];  // B operand: columns = i
            }
#pragma unroll
            for (int fi = 0; fi < 4; ++fi)
#pragma unroll
                for (int fj = 0; fj < 4; ++fj) acc[fi][fj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[fj], bv[fi], acc[fi][fj], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            cov_store_step(lds_i, tid, vi, valid);
            cov_store_step(lds_j, tid, vj, valid >> 4);
            __syncthreads();
        }
    }

    // ---- epilogue: scale, clip, NaN -> 0 in registers; then through this wave's LDS buffer, 32 output rows at a time ----
    const float inv_n1 = 1.0f / (float)(S - 1);  // torch.cov: unbiased (correction = 1)
    const int ib = i0 + wi * 64, jb = j0 + wj * 64;
#pragma unroll
    for (int fi = 0; fi < 4; ++fi) {
        const int i = ib + fi * 16 + (lane & 15);
        const float si = (use_cov || i >= P) ? 1.0f : inv_std[(size_t)b * P + i];
#pragma unroll
        for (int fj = 0; fj < 4; ++fj) {
            const int j = jb + fj * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float c = acc[fi][fj][r] * inv_n1;
                if (!use_cov && j + r < P) {
                    c = c * (si * inv_std[(size_t)b * P + j + r]);  // (s_i s_j first: the same bits for (i, j) and (j, i) -- the matrix is exactly symmetric)
                    // torch.corrcoef clips to [-1, 1] and keeps a NaN (a constant row: 0 * inf) a NaN; fminf / fmaxf would turn it into -1
                    if (c == c) c = fminf(1.0f, fmaxf(-1.0f, c));
                }
                acc[fi][fj][r] = (c != c) ? 0.f : c;  // NaN -> 0 (segmentation.py:541)
            }
        }
    }
    float* buf = lds + wave * (32 * kCovEpiPitch);  // 4 x 32 x 68 floats = 34816 B of the 36864-B operand area (every wave is past its last fragment read: barrier above)
    const int i_end = min(P, row0 + nrows);
    const bool row_vec = (P & 3) == 0;
    // orient 0: rows = i, columns = j (the tile itself); orient 1 (SYM, off-diagonal tiles): rows = j, columns = i (its mirror image)
    const int n_orient = (SYM && ti != tj) ? 2 : 1;
    for (int orient = 0; orient < n_orient; ++orient) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (orient == 0) {
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int fj = 0; fj < 4; ++fj)
                        *reinterpret_cast<f32x4*>(buf + (f * 16 + (lane & 15)) * kCovEpiPitch + fj * 16 + (lane >> 4) * 4) = acc[half * 2 + f][fj];
            } else {
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int fi = 0; fi < 4; ++fi)
#pragma unroll
                        for (int r = 0; r < 4; ++r) buf[(f * 16 + (lane >> 4) * 4 + r) * kCovEpiPitch + fi * 16 + (lane & 15)] = acc[fi][half * 2 + f][r];
            }
            // (the buffer is this wave's own and a wave's LDS operations execute in order: no workgroup barrier between its writes and its reads -- with eight
            // `__syncthreads()` per tile the epilogue cost ~100 us of the launch even with its stores removed; only the compiler must not move them across)
            __builtin_amdgcn_wave_barrier();
            const int rbase = (orient == 0 ? ib : jb) + half * 32, cbase = orient == 0 ? jb : ib;
            const int r_end = orient == 0 ? i_end : P, r_off = orient == 0 ? row0 : 0;  // (the mirror image exists only when the call covers every row)
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const int idx = lane + 64 * n, rr = idx >> 4, c4 = (idx & 15) * 4;
                const int row = rbase + rr, col = cbase + c4;
                if (row < r_end && col < P) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(buf + rr * kCovEpiPitch + c4);
                    float* dst = out + ((size_t)b * nrows + (row - r_off)) * P + col;
                    if (row_vec) {
                        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));  // (629 MB at P = 12544, written once: a streaming store)
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (col + r < P) dst[r] = v[r];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
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

// per column (b, s): {min, max, mean, unbiased std} over the P positions -> st[b][s].
// Round 6: the positions are cut into chunks, one workgroup per (64 columns, chunk) -- until round 5 ONE workgroup walked all P = 12544 positions of its 64
// columns twice (4 workgroups for S = 256: 1.1 ms for a 12.8-MB matrix, profiles/r6_bench_flowstats_before.json).  Every thread keeps a float64 Welford state
// (count, mean, M2) over its rows -- one pass, no cancellation --; the four row groups of a workgroup and then the chunks are merged with Chan's formula in a
// FIXED order (deterministic).  A NaN is tracked explicitly (torch.amin / amax propagate it; a NaN-free column holding +inf and -inf has a NaN mean but
// min = -inf, max = +inf).
struct ColPartial {
    double n, mean, m2;
    float mn, mx;
    int nan;
    int pad;
};

__device__ __forceinline__ void col_merge(double& n, double& mean, double& m2, double nb, double meanb, double m2b) {
    if (nb == 0.0) return;
    if (n == 0.0) {
        n = nb; mean = meanb; m2 = m2b;
        return;
    }
    const double tot = n + nb, d = meanb - mean;
    mean += d * (nb / tot);
    m2 += m2b + d * d * (n * nb / tot);
    n = tot;
}

constexpr int kColChunkRows = 112;  // positions per workgroup (P = 12544: 112 chunks x S / 64 column groups)

__global__ __launch_bounds__(256) void flow_colpartial_kernel(const float* __restrict__ x, int P, int S, int n_chunks, ColPartial* __restrict__ part) {
    __shared__ double rn[4][64], rmean[4][64], rm2[4][64];
    __shared__ float rmn[4][64], rmx[4][64];
    __shared__ int rnan[4][64];
    const int b = blockIdx.z, chunk = blockIdx.y, c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int s = blockIdx.x * 64 + c;
    const bool in = s < S;
    const int p0 = chunk * kColChunkRows, p1 = min(P, p0 + kColChunkRows);
    const float* xb = x + (size_t)b * P * S + (in ? s : 0);
    // float64 sums of (v - K) and (v - K)^2 about the thread's first value K: no division per element, and the shift keeps M2 = sum2 - sum^2 / n free of
    // cancellation at float64 (the values of a column are within a few orders of magnitude of each other)
    double n = 0.0, mean = 0.0, m2 = 0.0;
    float mn = INFINITY, mx = -INFINITY;
    int has_nan = 0;
    if (in && p0 + rg < p1) {
        const double K = (double)xb[(size_t)(p0 + rg) * S];
        double sum = 0.0, sum2 = 0.0;
        for (int p = p0 + rg; p < p1; p += 4) {
            const float v = xb[(size_t)p * S];
            has_nan |= (v != v) ? 1 : 0;
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
            n += 1.0;
            const double d = (double)v - K;
            sum += d;
            sum2 += d * d;
        }
        mean = K + sum / n;
        m2 = sum2 - sum * sum / n;
        if (m2 < 0.0) m2 = 0.0;
    }
    rn[rg][c] = n; rmean[rg][c] = mean; rm2[rg][c] = m2;
    rmn[rg][c] = mn; rmx[rg][c] = mx; rnan[rg][c] = has_nan;
    __syncthreads();
    if (rg == 0 && in) {
        for (int g = 1; g < 4; ++g) {
            col_merge(n, mean, m2, rn[g][c], rmean[g][c], rm2[g][c]);
            mn = fminf(mn, rmn[g][c]);
            mx = fmaxf(mx, rmx[g][c]);
            has_nan |= rnan[g][c];
        }
        ColPartial o;
        o.n = n; o.mean = mean; o.m2 = m2; o.mn = mn; o.mx = mx; o.nan = has_nan; o.pad = 0;
        part[((size_t)b * n_chunks + chunk) * S + s] = o;
    }
}

// 64 columns x 16 chunk groups per workgroup: group g merges its chunks in ascending order, then group 0 merges the 16 group results in ascending order -- the
// same association for every launch (deterministic).  (One thread per column walking all 112 chunks: 43 us of dependent loads, profiles/r6_kernel_stats_bench_flowstats*.csv.)
__global__ __launch_bounds__(1024) void flow_colfinish_kernel(const ColPartial* __restrict__ part, int P, int S, int n_chunks, float4* __restrict__ st) {
    __shared__ double gn[16][64], gmean[16][64], gm2[16][64];
    __shared__ float gmn[16][64], gmx[16][64];
    __shared__ int gnan[16][64];
    const int b = blockIdx.y, c = threadIdx.x & 63, g = threadIdx.x >> 6, s = blockIdx.x * 64 + c;
    const int per = (n_chunks + 15) / 16;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    float mn = INFINITY, mx = -INFINITY;
    int has_nan = 0;
    if (s < S)
        for (int ch = g * per; ch < min(n_chunks, (g + 1) * per); ++ch) {
            const ColPartial q = part[((size_t)b * n_chunks + ch) * S + s];
            col_merge(n, mean, m2, q.n, q.mean, q.m2);
            mn = fminf(mn, q.mn);
            mx = fmaxf(mx, q.mx);
            has_nan |= q.nan;
        }
    gn[g][c] = n; gmean[g][c] = mean; gm2[g][c] = m2;
    gmn[g][c] = mn; gmx[g][c] = mx; gnan[g][c] = has_nan;
    __syncthreads();
    if (g != 0 || s >= S) return;
    for (int k = 1; k < 16; ++k) {
        col_merge(n, mean, m2, gn[k][c], gmean[k][c], gm2[k][c]);
        mn = fminf(mn, gmn[k][c]);
        mx = fmaxf(mx, gmx[k][c]);
        has_nan |= gnan[k][c];
    }
    const double var = m2 / (double)(P - 1);  // P == 1: 0 / 0 = NaN, as torch.std
    const float qn = __builtin_nanf("");
    st[(size_t)b * S + s] = make_float4(has_nan ? qn : mn, has_nan ? qn : mx, (float)mean, (float)sqrt(var));
}

// elementwise forms; op: 1 x * (x > a), 2 (x > a), 3 ((x - min) > a * (max - min)), 4 x / max(colmax, eps), 5 (x - mean) / max(std, eps)
// grid (column groups of 64, row blocks, B): a thread keeps its column's statistics in registers and walks 16 positions
constexpr int kApplyRows = 16;
__global__ __launch_bounds__(256) void flow_apply_kernel(float* __restrict__ x, int P, int S, int op, float a, float eps, const float4* __restrict__ st) {
    const int b = blockIdx.z, s = blockIdx.x * 64 + (threadIdx.x & 63);
    if (s >= S) return;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (op >= 3) c = st[(size_t)b * S + s];
    const int p0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * kApplyRows;
    float* xb = x + (size_t)b * P * S + s;
#pragma unroll 4
    for (int p = p0; p < min(P, p0 + kApplyRows); ++p) {
        const float v = xb[(size_t)p * S];
        float o;
        if (op == 1) o = v * ((v > a) ? 1.0f : 0.0f);
        else if (op == 2) o = (v > a) ? 1.0f : 0.0f;
        else if (op == 3) o = ((v - c.x) > a * (c.y - c.x)) ? 1.0f : 0.0f;
        else if (op == 4) o = v / (c.y != c.y ? c.y : fmaxf(c.y, eps));   // tensor.clamp(min=eps) keeps a NaN
        else o = (v - c.z) / (c.w != c.w ? c.w : fmaxf(c.w, eps));
        xb[(size_t)p * S] = o;
    }
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

// ---- the reference layout [B, C, H, W, S] with the sample axis innermost and (H, W, S) packed (sw == S, sh == W S, ss == 1): round 6 -----------------
// The kernels above give a thread one pixel (or one sample) and let it walk the other axis with a stride: every 4-byte load of a wave sits in a different
// cache line (S = 256: 388 us for 2 x 103 MB, profiles/r6_bench_flowstats_before.json).  For the packed layout a channel plane is ONE contiguous array of
// H W S floats: a workgroup takes 64 - 256 consecutive pixels (mag_tile_pix) = that many times S consecutive floats per channel, loads them with coalesced 4-byte loads
// (lanes along the sample axis), keeps the magnitudes in LDS [pixels][S + 1] and then reduces along whichever axis the pass needs:
//   pass 1 (per-sample range): thread s scans the pixels of its column -> one atomicMin / atomicMax per (workgroup, sample) on the float bits (magnitudes
//          are >= 0, where unsigned order = float order; min starts at 0xFFFFFFFF, max at 0);
//   pass 2 (the sum over the samples): thread p walks the S samples of its pixel IN SAMPLE ORDER -- the same additions in the same order as the strided kernel
//          above (bit-identical to it).
// The per-sample range is reduced with atomicMin / atomicMax on the float bits.  Every workgroup of a pass hits the same S addresses: 784 workgroups on 512 (or 48)
// addresses serialise at L2 (S = 24: 19.6 us of a 9.8-MB pass was the queue of 196 atomics per address).  So there are kRangeReplicas copies of the [B][S] arrays,
// workgroup g uses copy g mod kRangeReplicas, and the consumers fold the copies (min / max are exact in any order).
constexpr int kRangeReplicas = 8;

// pixels per workgroup: 256 for S <= 32, 128 for S <= 64, else 64 (mag_tile_pix): the tile stays <= 34 KB, and at S = 24 a quarter as many workgroups hit the S range
// atomics (784 workgroups x 24 samples on 48 addresses were the whole 24-us launch)
static inline int mag_tile_pix(int S) { return S <= 32 ? 256 : S <= 64 ? 128 : 64; }

__device__ __forceinline__ void mag_tile_to_lds(const float* __restrict__ f, int64_t sb, int64_t sc, int b, int C, int HW, int S, int pix0, int tile_pix, float* mag) {
    const int npix = min(tile_pix, HW - pix0);
    const int n = npix * S;
    const float* fb = f + b * sb + (int64_t)pix0 * S;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        float a = 0.f;
        for (int c = 0; c < C; ++c) {
            const float v = fb[c * sc + e];
            a += v * v;
        }
        const int pl = e / S;
        mag[pl * (S + 1) + (e - pl * S)] = sqrtf(a);
    }
}

__global__ __launch_bounds__(256) void flow_mag_minmax_packed_kernel(const float* __restrict__ f, int64_t sb, int64_t sc, int C, int HW, int S, int tile_pix,
                                                                     unsigned* __restrict__ mn_bits, unsigned* __restrict__ mx_bits) {
    extern __shared__ float mag[];  // [tile_pix][S + 1]
    const int b = blockIdx.y, pix0 = blockIdx.x * tile_pix;
    mag_tile_to_lds(f, sb, sc, b, C, HW, S, pix0, tile_pix, mag);
    __syncthreads();
    const int npix = min(tile_pix, HW - pix0);
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        float mn = INFINITY, mx = -INFINITY;
        for (int pl = 0; pl < npix; ++pl) {
            const float m = mag[pl * (S + 1) + s];
            mn = fminf(mn, m);
            mx = fmaxf(mx, m);
        }
        const size_t rep = (size_t)(blockIdx.x % kRangeReplicas) * gridDim.y * S;
        if (mn == mn && mn != INFINITY) atomicMin(&mn_bits[rep + (size_t)b * S + s], __float_as_uint(mn));
        if (mx == mx && mx != -INFINITY) atomicMax(&mx_bits[rep + (size_t)b * S + s], __float_as_uint(mx));
    }
}

__global__ __launch_bounds__(256) void flow_motion_sum_packed_kernel(const float* __restrict__ f, int64_t sb, int64_t sc, int C, int HW, int S, int tile_pix,
                                                                     const unsigned* __restrict__ mn_bits, const unsigned* __restrict__ mx_bits, float eps,
                                                                     float* __restrict__ sum) {
    extern __shared__ float mag[];  // [tile_pix][S + 1]
    const int b = blockIdx.y, pix0 = blockIdx.x * tile_pix;
    mag_tile_to_lds(f, sb, sc, b, C, HW, S, pix0, tile_pix, mag);
    float* lo_s = mag + (size_t)tile_pix * (S + 1);  // [S] range start, [S] range length: folded from the replicas once per workgroup
    float* len_s = lo_s + S;
    if (mn_bits)
        for (int s = threadIdx.x; s < S; s += blockDim.x) {
            unsigned l = 0xFFFFFFFFu, h = 0u;
            for (int r = 0; r < kRangeReplicas; ++r) {
                l = min(l, mn_bits[((size_t)r * gridDim.y + b) * S + s]);
                h = max(h, mx_bits[((size_t)r * gridDim.y + b) * S + s]);
            }
            const float lo = __uint_as_float(l);
            lo_s[s] = lo;
            len_s[s] = fmaxf(__uint_as_float(h) - lo, eps);
        }
    __syncthreads();
    const int npix = min(tile_pix, HW - pix0);
    if ((int)threadIdx.x >= npix) return;
    const float* row = mag + threadIdx.x * (S + 1);
    float acc = 0.f;
    for (int s = 0; s < S; ++s) {
        float m = row[s];
        if (mn_bits) m = (m - lo_s[s]) / len_s[s];
        acc += m;
    }
    sum[(size_t)b * HW + pix0 + threadIdx.x] = acc;
}

// ---- the same two passes for S = 64, 128 or a multiple of 256 (the counterfactual batch of BASELINE configs[3]: S = 256): no LDS tile at all --------------
// Q lanes own one pixel's samples as 16-byte loads (Q = 16 / 32 / 64 lanes x 4 samples; S = 256 k: k such loads per pixel): a pixel's channel values arrive as
// ONE coalesced 1-KB wave load per channel, the per-sample range sits in the lane's registers for the whole launch (the lane's samples never change), and the sum
// over the samples is a DPP / shuffle reduction over the Q lanes.  Every wave walks 16 pixels, four at a time (8 independent loads in flight per lane).
// (The LDS-tile form above at S = 256: 64 x 257 x 4 B = 66 KB per workgroup -> two workgroups per CU, and a 64-thread serial tail: 253 us for 2 x 103 MB.)
constexpr int kRowPixPerWave = 16;

template <int Q>
__device__ __forceinline__ float group_sum(float v) {  // sum over aligned groups of Q lanes; every lane of the group gets it
#pragma unroll
    for (int off = 1; off < Q; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int Q>
__global__ __launch_bounds__(256) void flow_mag_minmax_rows_kernel(const float* __restrict__ f, int64_t sb, int64_t sc, int C, int HW, int S,
                                                                   unsigned* __restrict__ mn_bits, unsigned* __restrict__ mx_bits) {
    __shared__ float red_mn[4][256], red_mx[4][256];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PPL = 64 / Q;                  // pixels per wave load
    const int sub = lane / Q, s4 = (lane % Q) * 4;
    const int chunks = Q == 64 ? S / 256 : 1;    // 256-sample pieces per pixel
    const int pix0 = (blockIdx.x * 4 + wave) * kRowPixPerWave;
    const float* fb = f + b * sb;
    for (int ch = 0; ch < chunks; ++ch) {
        f32x4 mn = f32x4{INFINITY, INFINITY, INFINITY, INFINITY}, mx = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll 4
        for (int i = 0; i < kRowPixPerWave; i += PPL) {
            const int pix = pix0 + i + sub;
            if (pix < HW) {
                f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int c = 0; c < C; ++c) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(fb + c * sc + (int64_t)pix * S + ch * 256 + s4);
                    a += v * v;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float m = sqrtf(a[r]);
                    mn[r] = fminf(mn[r], m);
                    mx[r] = fmaxf(mx[r], m);
                }
            }
        }
        // lanes that hold the same samples for other pixels of the wave load (Q < 64), then the four waves: LDS; then one atomic pair per sample and workgroup
#pragma unroll
        for (int off = Q; off < 64; off <<= 1)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mn[r] = fminf(mn[r], __shfl_xor(mn[r], off, 64));
                mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], off, 64));
            }
        if (lane < Q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                red_mn[wave][s4 + r] = mn[r];
                red_mx[wave][s4 + r] = mx[r];
            }
        __syncthreads();
        const int ns = Q * 4;  // samples of this piece
        for (int s = threadIdx.x; s < ns; s += 256) {
            const float lo = fminf(fminf(red_mn[0][s], red_mn[1][s]), fminf(red_mn[2][s], red_mn[3][s]));
            const float hi = fmaxf(fmaxf(red_mx[0][s], red_mx[1][s]), fmaxf(red_mx[2][s], red_mx[3][s]));
            const size_t rep = (size_t)(blockIdx.x % kRangeReplicas) * gridDim.y * S;
            if (lo != INFINITY) atomicMin(&mn_bits[rep + (size_t)b * S + ch * 256 + s], __float_as_uint(lo));
            if (hi != -INFINITY) atomicMax(&mx_bits[rep + (size_t)b * S + ch * 256 + s], __float_as_uint(hi));
        }
        __syncthreads();
    }
}

template <int Q>
__global__ __launch_bounds__(256) void flow_motion_sum_rows_kernel(const float* __restrict__ f, int64_t sb, int64_t sc, int C, int HW, int S,
                                                                   const unsigned* __restrict__ mn_bits, const unsigned* __restrict__ mx_bits, float eps,
                                                                   float* __restrict__ sum) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PPL = 64 / Q;
    const int sub = lane / Q, s4 = (lane % Q) * 4;
    const int chunks = Q == 64 ? S / 256 : 1;
    const int pix0 = (blockIdx.x * 4 + wave) * kRowPixPerWave;
    const float* fb = f + b * sb;
    auto range_of = [&](int ch, f32x4& lo, f32x4& range) {
        lo = f32x4{0.f, 0.f, 0.f, 0.f};
        range = f32x4{1.f, 1.f, 1.f, 1.f};
        if (mn_bits) {
            u32x4 l = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, h = u32x4{0u, 0u, 0u, 0u};
            for (int rep = 0; rep < kRangeReplicas; ++rep) {
                const size_t o = ((size_t)rep * gridDim.y + b) * S + ch * 256 + s4;
                const u32x4 l2 = *reinterpret_cast<const u32x4*>(mn_bits + o), h2 = *reinterpret_cast<const u32x4*>(mx_bits + o);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    l[r] = min(l[r], l2[r]);
                    h[r] = max(h[r], h2[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                lo[r] = __uint_as_float(l[r]);
                range[r] = fmaxf(__uint_as_float(h[r]) - lo[r], eps);
            }
        }
    };
    auto piece = [&](int pix, int ch, const f32x4& lo, const f32x4& range) {
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < C; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(fb + c * sc + (int64_t)pix * S + ch * 256 + s4);
            a += v * v;
        }
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float m = sqrtf(a[r]);
            if (mn_bits) m = (m - lo[r]) / range[r];
            part += m;
        }
        return part;
    };
    f32x4 lo0, range0;
    range_of(0, lo0, range0);  // (one piece per pixel -- S <= 256 -- : the lane's ranges stay in registers for the whole launch)
#pragma unroll 4
    for (int i = 0; i < kRowPixPerWave; i += PPL) {
        const int pix = pix0 + i + sub;
        float part = 0.f;
        if (pix < HW) {
            part = piece(pix, 0, lo0, range0);
            for (int ch = 1; ch < chunks; ++ch) {
                f32x4 lo, range;
                range_of(ch, lo, range);
                part += piece(pix, ch, lo, range);
            }
        }
        const float tot = group_sum<Q>(part);
        if ((lane % Q) == 0 && pix < HW) sum[(size_t)b * HW + pix] = tot;
    }
}

// map = sum * scale; if normalize: (map - min) / max(max - min, eps) over (H, W) per b.  One workgroup of 1024 threads per b.
// Maps of up to 65536 pixels whose size is a multiple of 4 (224 x 224 = 50176) stay in registers between the range pass and the normalisation: every thread issues its
// <= 16 independent 16-byte loads at once, one pass over memory (round 6; the two-pass loop below took 25 us for 200 KB -- a chain of 49 dependent 4-byte loads per thread, twice).
__global__ __launch_bounds__(1024) void flow_map_finish_kernel(float* __restrict__ map, int HW, float scale, int normalize, float eps) {
    __shared__ float red[32];
    float* m = map + (size_t)blockIdx.x * HW;
    float mn = INFINITY, mx = -INFINITY;
    if ((HW & 3) == 0 && HW <= 1024 * 4 * 16 && (((uintptr_t)m) & 15) == 0) {
        f32x4 v[16];
        const int n4 = HW / 4;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int e = threadIdx.x + 1024 * k;
            if (e < n4) v[k] = *reinterpret_cast<const f32x4*>(m + 4 * (size_t)e);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int e = threadIdx.x + 1024 * k;
            if (e < n4) {
                v[k] = v[k] * scale;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    mn = fminf(mn, v[k][r]);
                    mx = fmaxf(mx, v[k][r]);
                }
            }
        }
        float d = 1.f;
        if (normalize) {
            block_minmax(mn, mx, red);
            d = fmaxf(mx - mn, eps);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int e = threadIdx.x + 1024 * k;
            if (e < n4) {
                if (normalize)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[k][r] = (v[k][r] - mn) / d;
                *reinterpret_cast<f32x4*>(m + 4 * (size_t)e) = v[k];
            }
        }
        return;
    }
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
    if (strides[4] == 1 && S % 4 == 0 && ((uintptr_t)flows_dev & 15) == 0 && ((uintptr_t)x_dev & 15) == 0 && strides[0] % 4 == 0 && strides[1] % 4 == 0 && strides[2] % 4 == 0 &&
        strides[3] % 4 == 0) {
        const int64_t total4 = (int64_t)B * (H / downsample) * (W / downsample) * (S / 4);
        hipLaunchKernelGGL(flow_features4_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flows_dev, strides[0], strides[1], strides[2],
                           strides[3], B, C, H, W, S, downsample, x_dev);
        CWM_HIP_CHECK(hipGetLastError());
        return 0;
    }
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
    const int T = (P + 127) / 128;
    const bool vec = S % 4 == 0 && ((uintptr_t)xc_work_dev & 15) == 0;
    const bool sym = row0 == 0 && nrows == P;  // the whole (symmetric) matrix: tiles on and above the diagonal, each off-diagonal one written twice
    const int mode = !vec ? 0 : (P % 128 == 0 && S % kCovBK == 0 && row0 % 128 == 0) ? 2 : 1;
    const int uc = use_covariance ? 1 : 0;
    const dim3 grid = sym ? dim3((unsigned)(T * (T + 1) / 2), 1, (unsigned)B) : dim3((unsigned)T, (unsigned)((nrows + 127) / 128), (unsigned)B);
#define CWM_COV_LAUNCH(SYM_, MODE_) \
    hipLaunchKernelGGL((flow_cov_kernel<SYM_, MODE_>), grid, dim3(256), 0, s, xc_work_dev, inv_std_work_dev, P, S, row0, nrows, uc, out_dev)
    if (sym) {
        if (mode == 2) CWM_COV_LAUNCH(true, 2);
        else if (mode == 1) CWM_COV_LAUNCH(true, 1);
        else CWM_COV_LAUNCH(true, 0);
    } else {
        if (mode == 2) CWM_COV_LAUNCH(false, 2);
        else if (mode == 1) CWM_COV_LAUNCH(false, 1);
        else CWM_COV_LAUNCH(false, 0);
    }
#undef CWM_COV_LAUNCH
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int cwm_flow_transform(float* x_dev, int B, int P, int S, int spearman, int thresh_mode, float thresh, int normalize, int zscore, float eps,
                                  float* stats_work_dev, void* stream) {
    CWM_REQUIRE(x_dev && B > 0 && P > 0 && S > 0, "cwm_flow_transform: bad argument");
    CWM_REQUIRE(thresh_mode >= 0 && thresh_mode <= 3, "cwm_flow_transform: thresh_mode must be 0 (none), 1 (x * (x > t)), 2 (x > t) or 3 (range threshold)");
    CWM_REQUIRE(!(thresh_mode == 3 || normalize || zscore) || stats_work_dev, "cwm_flow_transform: the column statistics need the work buffer (cwm_flow_transform_work_bytes)");
    CWM_REQUIRE(((uintptr_t)stats_work_dev & 15) == 0, "cwm_flow_transform: the work buffer must be 16-byte aligned");
    CWM_REQUIRE(!spearman || S <= 4096, "cwm_flow_transform: Spearman ranks support at most 4096 samples (S=%d)", S);
    hipStream_t s = (hipStream_t)stream;
    float4* st = reinterpret_cast<float4*>(stats_work_dev);
    // work buffer (cwm_flow_transform_work_bytes): [B][S] float4 statistics, then the [B][n_chunks][S] partials of the column pass
    const int n_chunks = (P + kColChunkRows - 1) / kColChunkRows;
    ColPartial* part = stats_work_dev ? reinterpret_cast<ColPartial*>(stats_work_dev + (size_t)4 * B * S) : nullptr;
    const dim3 pgrid((unsigned)((S + 63) / 64), (unsigned)n_chunks, (unsigned)B), fgrid((unsigned)((S + 63) / 64), (unsigned)B);
    const dim3 agrid((unsigned)((S + 63) / 64), (unsigned)((P + 4 * kApplyRows - 1) / (4 * kApplyRows)), (unsigned)B);
    auto colstats = [&]() {
        hipLaunchKernelGGL(flow_colpartial_kernel, pgrid, dim3(256), 0, s, x_dev, P, S, n_chunks, part);
        hipLaunchKernelGGL(flow_colfinish_kernel, fgrid, dim3(1024), 0, s, part, P, S, n_chunks, st);
    };
    if (spearman) {
        const size_t smem = (size_t)4 * S * sizeof(float);
        if (smem > 48 * 1024)
            if (int rc = cwm_set_max_lds((const void*)flow_argsort_rows_kernel, (int)smem)) return rc;
        hipLaunchKernelGGL(flow_argsort_rows_kernel, dim3((unsigned)((B * P + 3) / 4)), dim3(256), smem, s, x_dev, B * P, S);
    }
    if (thresh_mode == 1 || thresh_mode == 2) {
        hipLaunchKernelGGL(flow_apply_kernel, agrid, dim3(256), 0, s, x_dev, P, S, thresh_mode, thresh, eps, st);
    } else if (thresh_mode == 3) {
        colstats();
        hipLaunchKernelGGL(flow_apply_kernel, agrid, dim3(256), 0, s, x_dev, P, S, 3, thresh, eps, st);
    }
    if (normalize) {
        colstats();
        hipLaunchKernelGGL(flow_apply_kernel, agrid, dim3(256), 0, s, x_dev, P, S, 4, 0.f, eps, st);
    }
    if (zscore) {
        colstats();
        hipLaunchKernelGGL(flow_apply_kernel, agrid, dim3(256), 0, s, x_dev, P, S, 5, 0.f, eps, st);
    }
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" size_t cwm_flow_transform_work_bytes(int B, int P, int S) {
    if (B <= 0 || P <= 0 || S <= 0) return 0;
    const size_t n_chunks = (size_t)(P + kColChunkRows - 1) / kColChunkRows;
    return (size_t)16 * B * S + sizeof(ColPartial) * (size_t)B * n_chunks * S;
}

extern "C" size_t cwm_flow_motion_work_bytes(int B, int S) {
    if (B <= 0 || S <= 0) return 0;
    return (size_t)2 * kRangeReplicas * B * S * sizeof(float);
}

extern "C" int cwm_flow_motion_sum(const float* flows_dev, const int64_t* strides, int B, int C, int H, int W, int S, int normalize_per_sample,
                                   float eps, float* minmax_work_dev, float* sum_dev, void* stream) {
    CWM_REQUIRE(flows_dev && strides && sum_dev && B > 0 && C > 0 && H > 0 && W > 0 && S > 0, "cwm_flow_motion_sum: bad argument");
    CWM_REQUIRE(!normalize_per_sample || minmax_work_dev, "cwm_flow_motion_sum: per-sample normalisation needs the work buffer (cwm_flow_motion_work_bytes)");
    hipStream_t s = (hipStream_t)stream;
    // the reference's layout (sample axis innermost, (H, W, S) packed): the coalesced kernels; any other strides: the strided ones
    const int tile_mm = mag_tile_pix(S), tile_sum = 64;  // (the range pass wants few workgroups -- fewer atomics --, the sum pass many)
    const size_t smem_mm = (size_t)tile_mm * (S + 1) * sizeof(float);
    const size_t smem_sum = ((size_t)tile_sum * (S + 1) + 2 * (size_t)S) * sizeof(float);
    const bool packed = strides[4] == 1 && strides[3] == S && strides[2] == (int64_t)W * S;
    const int q = S == 64 ? 16 : S == 128 ? 32 : (S % 256 == 0 ? 64 : 0);  // lanes per pixel of the register form (0: the LDS-tile form)
    const bool rows_form = packed && q && strides[0] % 4 == 0 && strides[1] % 4 == 0 && ((uintptr_t)flows_dev & 15) == 0 &&
                           (!normalize_per_sample || ((uintptr_t)minmax_work_dev & 15) == 0);
    const bool tile_form = !rows_form && packed && smem_mm <= 150 * 1024 && smem_sum <= 150 * 1024;
    if (rows_form || tile_form) {
        const int HW = H * W;
        // work buffer (cwm_flow_motion_work_bytes): kRangeReplicas x [B][S] min bits, then as many max bits
        unsigned *mn = nullptr, *mx = nullptr;
        if (normalize_per_sample) {
            const size_t n = (size_t)kRangeReplicas * B * S;
            mn = reinterpret_cast<unsigned*>(minmax_work_dev);
            mx = mn + n;
            CWM_HIP_CHECK(hipMemsetAsync(mn, 0xFF, n * sizeof(unsigned), s));
            CWM_HIP_CHECK(hipMemsetAsync(mx, 0x00, n * sizeof(unsigned), s));
        }
        if (rows_form) {
            const dim3 grid((unsigned)((HW + 4 * kRowPixPerWave - 1) / (4 * kRowPixPerWave)), (unsigned)B);
            if (normalize_per_sample) {
                if (q == 16) hipLaunchKernelGGL(flow_mag_minmax_rows_kernel<16>, grid, dim3(256), 0, s, flows_dev, strides[0], strides[1], C, HW, S, mn, mx);
                else if (q == 32) hipLaunchKernelGGL(flow_mag_minmax_rows_kernel<32>, grid, dim3(256), 0, s, flows_dev, strides[0], strides[1], C, HW, S, mn, mx);
                else hipLaunchKernelGGL(flow_mag_minmax_rows_kernel<64>, grid, dim3(256), 0, s, flows_dev, strides[0], strides[1], C, HW, S, mn, mx);
            }
            if (q == 16) hipLaunchKernelGGL(flow_motion_sum_rows_kernel<16>, grid, dim3(256), 0, s, flows_dev, strides[0], strides[1], C, HW, S, mn, mx, eps, sum_dev);
            else if (q == 32) hipLaunchKernelGGL(flow_motion_sum_rows_kernel<32>, grid, dim3(256), 0, s, flows_dev, strides[0], strides[1], C, HW, S, mn, mx, eps, sum_dev);
            else hipLaunchKernelGGL(flow_motion_sum_rows_kernel<64>, grid, dim3(256), 0, s, flows_dev, strides[0], strides[1], C, HW, S, mn, mx, eps, sum_dev);
        } else {
            if (smem_mm > 48 * 1024)
                if (int rc = cwm_set_max_lds((const void*)flow_mag_minmax_packed_kernel, (int)smem_mm)) return rc;
            if (smem_sum > 48 * 1024)
                if (int rc = cwm_set_max_lds((const void*)flow_motion_sum_packed_kernel, (int)smem_sum)) return rc;
            if (normalize_per_sample)
                hipLaunchKernelGGL(flow_mag_minmax_packed_kernel, dim3((unsigned)((HW + tile_mm - 1) / tile_mm), (unsigned)B), dim3(256), smem_mm, s, flows_dev, strides[0],
                                   strides[1], C, HW, S, tile_mm, mn, mx);
            hipLaunchKernelGGL(flow_motion_sum_packed_kernel, dim3((unsigned)((HW + tile_sum - 1) / tile_sum), (unsigned)B), dim3(256), smem_sum, s, flows_dev, strides[0],
                               strides[1], C, HW, S, tile_sum, mn, mx, eps, sum_dev);
        }
        CWM_HIP_CHECK(hipGetLastError());
        return 0;
    }
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
