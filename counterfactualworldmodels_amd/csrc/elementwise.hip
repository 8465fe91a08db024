// HBM-bound edge kernels of the VMAE predictor path (gfx950): LayerNorm, mask -> token permutation
// (bit-exact index op), frame load + imagenet-normalise + tubelet patch gather, decoder mask-token
// fill, patch un-embed scatter, fp32 -> bf16 (hi, lo) split.
#include "common.h"
#include "kernels.h"
#include <algorithm>

namespace cwm {

// ---------------------------------------------------------------------------------------------
// LayerNorm (eps 1e-6, affine) -> bf16 hi(/lo) planes.  Reference: nn.LayerNorm call sites
// VideoMAE/utils.py:148-149 (norm1/norm2), vmae.py:172 (encoder.norm), :251 (decoder.norm on the
// last Nm tokens).  One wave per row, row kept in registers (16-byte loads and stores), two-pass mean/variance in fp32.
// ---------------------------------------------------------------------------------------------
template <int PLANES>
__global__ __launch_bounds__(256) void layernorm_kernel(const LayerNormParams p) {
    constexpr int MAXI = 2;  // D <= 1024: lane l owns elements 8 l + 512 i .. + 7 (two 16-byte loads; ONE 16-byte store per plane: round 4 --
                             // 8-byte stores ran at 0.55-0.7 of the 16-byte rate, MI355X_MICROARCH.md; 8 consecutive k never straddle a
                             // 32-k [hi | lo] block)
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.rows) return;
    int in_row = r;
    if (p.rows_out_per_b > 0) {
        const int b = r / p.rows_out_per_b;
        in_row = b * p.rows_in_per_b + p.in_offset + (r - b * p.rows_out_per_b);
    }
    const float* x = p.x + (size_t)in_row * p.ldx;
    f32x4 v[MAXI][2];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int idx = lane * 8 + i * 512;
        if (idx < p.D) {
            v[i][0] = *reinterpret_cast<const f32x4*>(x + idx);
            v[i][1] = *reinterpret_cast<const f32x4*>(x + idx + 4);
            s += ((v[i][0][0] + v[i][0][1]) + (v[i][0][2] + v[i][0][3])) + ((v[i][1][0] + v[i][1][1]) + (v[i][1][2] + v[i][1][3]));
        } else {
            v[i][0] = v[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float mean = wave_sum_dpp(s) / (float)p.D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int idx = lane * 8 + i * 512;
        if (idx < p.D) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = v[i][h][e] - mean;
                    sq += a * a;
                }
        }
    }
    const float rstd = rsqrtf(wave_sum_dpp(sq) / (float)p.D + p.eps);
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int idx = lane * 8 + i * 512;
        if (idx < p.D) {
            bf16* out = p.out + a_pos<PLANES>(r, p.ldo, idx);
            bf16x8 hv, lv;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + idx + 4 * h);
                const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + idx + 4 * h);
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[e] = (v[i][h][e] - mean) * rstd * g[e] + be[e];
                    bf16 hh, ll;
                    split_bf16(y[e], hh, ll);
                    hv[4 * h + e] = hh;
                    lv[4 * h + e] = ll;
                }
                if (p.out_f32) *reinterpret_cast<f32x4*>(p.out_f32 + (size_t)r * p.D + idx + 4 * h) = y;
            }
            *reinterpret_cast<bf16x8*>(out) = hv;
            if constexpr (PLANES == 2) *reinterpret_cast<bf16x8*>(out + kLoOffset) = lv;
        }
    }
}

int launch_layernorm(const LayerNormParams& p, int planes, hipStream_t stream) {
    CWM_REQUIRE(p.D % 8 == 0 && p.D <= 1024, "layernorm: D=%d must be a multiple of 8 and <= 1024", p.D);
    CWM_REQUIRE(p.ldx % 4 == 0 && p.ldo % 8 == 0, "layernorm: row strides must be multiples of 4 (input) / 8 (output)");
    const int blocks = (p.rows + 3) / 4;
    if (planes == 1)
        hipLaunchKernelGGL(layernorm_kernel<1>, dim3(blocks), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL(layernorm_kernel<2>, dim3(blocks), dim3(256), 0, stream, p);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// mask -> permutation.  Reference: `x[~mask].reshape(B,-1,C)` (vmae.py:167) and the decoder order
// `cat([vis, masked])` (vmae.py:555-557): visible tokens in ascending token index, then masked
// tokens in ascending token index.  Integer-exact; one workgroup per sample, block-wide scan.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_to_perm_kernel(const uint8_t* mask, int Nt, int n_vis, int* perm, int* err) {
    __shared__ int counts[256];
    __shared__ int total_vis;
    const int b = blockIdx.x, t = threadIdx.x;
    const uint8_t* m = mask + (size_t)b * Nt;
    const int per = (Nt + 255) / 256;
    const int lo = min(t * per, Nt), hi = min(lo + per, Nt);
    int c = 0;
    for (int i = lo; i < hi; ++i) c += (m[i] == 0);
    counts[t] = c;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) {
            const int v = counts[i];
            counts[i] = run;
            run += v;
        }
        total_vis = run;
        if (run != n_vis) atomicExch(err, 1);
    }
    __syncthreads();
    int vis_before = counts[t];
    const int tv = total_vis;
    int* pr = perm + (size_t)b * Nt;
    for (int i = lo; i < hi; ++i) {
        if (m[i] == 0) {
            pr[vis_before] = i;
            ++vis_before;
        } else {
            pr[tv + (i - vis_before)] = i;
        }
    }
}

int launch_mask_to_perm(const uint8_t* mask, int B, int Nt, int n_vis, int* perm, int* err, hipStream_t stream) {
    hipLaunchKernelGGL(mask_to_perm_kernel, dim3(B), dim3(256), 0, stream, mask, Nt, n_vis, perm, err);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void perm_to_rank_kernel(const int* perm, int* rank, int Nt, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int b = i / Nt;
    rank[(size_t)b * Nt + perm[i]] = i - b * Nt;
}

int launch_perm_to_rank(const int* perm, int* rank, int B, int Nt, hipStream_t stream) {
    const int total = B * Nt;
    hipLaunchKernelGGL(perm_to_rank_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, perm, rank, Nt, total);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Frame load + imagenet normalise + tubelet patch gather (im2col of the visible tokens only).
// Reference: `_preprocess` + `imagenet_normalize` (prediction.py:304-312, utils.py:15-21) and the
// Conv3d patch embed (VideoMAE/utils.py:174-197), whose GEMM is done by gemm.hip on this matrix.
// The reference embeds all Nt tokens and gathers afterwards (vmae.py:155-167); the embed is
// per-token, so gathering first is identical and skips the masked half of frame 2.
// One thread per (row, c, ph): reads P contiguous pixels, writes P contiguous bf16.
// ---------------------------------------------------------------------------------------------
template <int PLANES>
__global__ __launch_bounds__(256) void patch_gather_kernel(const PatchGatherParams p) {
    const int per_row = p.C * p.P;
    const int64_t total = (int64_t)p.B * p.n_rows * per_row;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int row = (int)(gid / per_row);
    const int rem = (int)(gid - (int64_t)row * per_row);
    const int c = rem / p.P, ph = rem - c * p.P;
    const int b = row / p.n_rows, i = row - b * p.n_rows;
    const int tau = p.perm[(size_t)b * (p.perm_stride ? p.perm_stride : p.Nt) + i];
    const int gw = p.W / p.P;
    const int n = (p.H / p.P) * gw;
    const int t = tau / n, hw = tau - t * n;
    const int hy = hw / gw, wx = hw - hy * gw;
    const float* src = p.x + b * p.sb + c * p.sc + t * p.st + (int64_t)(hy * p.P + ph) * p.W + wx * p.P;
    const float mean = (c == 0) ? 0.485f : (c == 1) ? 0.456f : 0.406f;
    const float stdv = (c == 0) ? 0.229f : (c == 1) ? 0.224f : 0.225f;
    const int kbase = c * p.P * p.P + ph * p.P;
    const bool pad_slot = tau >= p.Nt;  // null-token pad slot of a padded predictor: no pixels behind it
    for (int pw = 0; pw < p.P; pw += 4) {
        const float4 v = pad_slot ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(src + pw);
        float f[4] = {v.x, v.y, v.z, v.w};
        bf16x4 hv, lv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = f[e];
            if (p.normalize && !pad_slot) a = (a - mean) / stdv;
            bf16 hi, lo;
            split_bf16(a, hi, lo);
            hv[e] = hi;
            lv[e] = lo;
        }
        bf16* dst = p.out + a_pos<PLANES>(row, p.ld, kbase + pw);
        *reinterpret_cast<bf16x4*>(dst) = hv;
        if constexpr (PLANES == 2) *reinterpret_cast<bf16x4*>(dst + kLoOffset) = lv;
    }
    // zero the K padding (K = C*P*P rounded up to ld) once per row
    if (rem == 0) {
        const int K = p.C * p.P * p.P;
        for (int k = K; k < p.ld; ++k) {
            bf16* z = p.out + a_pos<PLANES>(row, p.ld, k);
            *z = (bf16)0.f;
            if constexpr (PLANES == 2) z[kLoOffset] = (bf16)0.f;
        }
    }
}

int launch_patch_gather(const PatchGatherParams& p, int planes, hipStream_t stream) {
    CWM_REQUIRE(p.P % 4 == 0 && p.W % 4 == 0, "patch_gather: patch size and width must be multiples of 4");
    CWM_REQUIRE(p.C == 3 || !p.normalize, "patch_gather: imagenet normalisation needs 3 channels");
    CWM_REQUIRE(p.ld >= p.C * p.P * p.P && p.ld % 4 == 0, "patch_gather: bad ld");
    const int64_t total = (int64_t)p.B * p.n_rows * p.C * p.P;
    const int blocks = (int)((total + 255) / 256);
    if (planes == 1)
        hipLaunchKernelGGL(patch_gather_kernel<1>, dim3(blocks), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL(patch_gather_kernel<2>, dim3(blocks), dim3(256), 0, stream, p);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// The index prologue of a forward in ONE launch (round 5): mask -> permutation (visible tokens ascending, then masked ascending:
// vmae.py:167, :555-557), its inverse `rank` (what the un-embed scatter reads, prediction.py:252-259), the per-row check of the visible
// count (the reference's reshape failure at vmae.py:167) and the patch gather above.  Until round 4 these were four dependent launches
// (memset of the error word, mask_to_perm, patch_gather, and perm_to_rank before the un-embed) of 4 - 12 us each with the platform's ~6 us
// between dependent kernels: ~25 us of the 1.86-ms batch-1 forward.  grid = (workgroups per sample, B).  Every workgroup of a sample
// scans the sample's mask row itself (1.5 - 6 KB from L2, a wave-shuffle prefix sum) into an LDS table {i-th visible token -> token index}
// and gathers its share of the visible rows from it; workgroup 0 of the sample also writes perm / rank / the row check.  Integer-exact.
// ---------------------------------------------------------------------------------------------
template <int PLANES>
__global__ __launch_bounds__(256) void index_gather_kernel(const PatchGatherParams p, const uint8_t* __restrict__ mask, int n_vis, int* __restrict__ perm_out,
                                                           int* __restrict__ rank_out, int* __restrict__ err_rows) {
    extern __shared__ int vis_tab[];  // [n_rows]
    __shared__ int wave_tot[4];
    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int L = p.perm_stride ? p.perm_stride : p.Nt;  // mask row length (padded predictors: real tokens + pad slots)
    const uint8_t* m = mask + (size_t)b * L;
    // perm / rank of the sample are written by ALL of its workgroups (every one has the whole scan anyway): thread slice t belongs to workgroup t mod nw.
    // Until round 5 workgroup 0 wrote all L entries alone: 25 dependent rounds of scattered 4-byte stores on the critical path of the launch.
    const int nw = min((int)gridDim.x, 256);
    const bool writer = (t % nw) == (int)blockIdx.x;
    int* pr = perm_out + (size_t)b * L;
    int* rk = rank_out ? rank_out + (size_t)b * L : nullptr;
    int total_vis;
    if ((L & 15) == 0 && L <= 256 * 32 && ((uintptr_t)mask & 15) == 0) {
        // Round 6: the thread's slice of the mask row -- 16 or 32 consecutive bytes -- comes in with one or two 16-byte loads and STAYS in registers for the
        // second pass.  (Round 5: (L + 255) / 256 = 25 single-byte loads per thread at a 25-byte lane stride, twice; on the long rows of ViT-L/4
        // -- L = 6272 -- that scan, repeated by each of the sample's ~150 workgroups, was most of the launch: 27 - 34 us for 10 MB, now 20 - 23.)
        const int per = L <= 256 * 16 ? 16 : 32;
        const int lo = t * per;
        u32x4 w0 = u32x4{0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}, w1 = w0;  // (past the row: "masked", never counted)
        if (lo < L) w0 = *reinterpret_cast<const u32x4*>(m + lo);
        if (per == 32 && lo + 16 < L) w1 = *reinterpret_cast<const u32x4*>(m + lo + 16);
        const int n_mine = lo >= L ? 0 : min(per, L - lo);
        auto zero_bytes = [](unsigned x) {  // number of bytes of x that are 0 (any non-zero byte = masked, as `m[i] == 0` read it)
            const unsigned y = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
            return __popc(~(y | x | 0x7F7F7F7Fu));
        };
        int c = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) c += zero_bytes(w0[q]) + zero_bytes(w1[q]);
        int incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int v = incl - c;
        for (int w2 = 0; w2 < wave; ++w2) v += wave_tot[w2];
        total_vis = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (k < n_mine) {
                const unsigned word = k < 16 ? w0[(k >> 2) & 3] : w1[(k >> 2) & 3];
                const bool visible = ((word >> (8 * (k & 3))) & 0xFFu) == 0;
                const int i = lo + k;
                if (visible) {
                    if (v < p.n_rows) vis_tab[v] = i;
                    if (writer) {
                        pr[v] = i;
                        if (rk) rk[i] = v;
                    }
                    ++v;
                } else if (writer) {
                    const int pos = total_vis + (i - v);
                    pr[pos] = i;
                    if (rk) rk[i] = pos;
                }
            }
        }
    } else {  // rows that are not whole 16-byte groups (test-sized grids): the byte loop
        const int per = (L + 255) / 256;
        const int lo = min(t * per, L), hi = min(lo + per, L);
        int c = 0;
        for (int i = lo; i < hi; ++i) c += (m[i] == 0);
        int incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int v = incl - c;
        for (int w2 = 0; w2 < wave; ++w2) v += wave_tot[w2];
        total_vis = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        for (int i = lo; i < hi; ++i) {
            if (m[i] == 0) {
                if (v < p.n_rows) vis_tab[v] = i;
                if (writer) {
                    pr[v] = i;
                    if (rk) rk[i] = v;
                }
                ++v;
            } else if (writer) {
                const int pos = total_vis + (i - v);
                pr[pos] = i;
                if (rk) rk[i] = pos;
            }
        }
    }
    if (blockIdx.x == 0 && t == 0) err_rows[b] = total_vis != n_vis ? 1 : 0;
    __syncthreads();

    const int per_row = p.C * p.P;
    const int total = p.n_rows * per_row;
    const int gw = p.W / p.P;
    const int n = (p.H / p.P) * gw;
    for (int gid = blockIdx.x * 256 + t; gid < total; gid += gridDim.x * 256) {
        const int i = gid / per_row;
        const int rem = gid - i * per_row;
        const int ch = rem / p.P, ph = rem - ch * p.P;
        const int tau = i < total_vis ? vis_tab[i] : p.Nt;  // (a row with too few visible tokens is reported through err_rows: its missing rows read nothing)
        const bool pad_slot = tau >= p.Nt;                  // null-token pad slot of a padded predictor: no pixels behind it
        const int tt = pad_slot ? 0 : tau;
        const int tf = tt / n, hw = tt - tf * n;
        const int hy = hw / gw, wx = hw - hy * gw;
        const float* src = p.x + b * p.sb + ch * p.sc + tf * p.st + (int64_t)(hy * p.P + ph) * p.W + wx * p.P;
        const float mean = (ch == 0) ? 0.485f : (ch == 1) ? 0.456f : 0.406f;
        const float stdv = (ch == 0) ? 0.229f : (ch == 1) ? 0.224f : 0.225f;
        const int kbase = ch * p.P * p.P + ph * p.P;
        const int64_t row = (int64_t)b * p.n_rows + i;
        for (int pw = 0; pw < p.P; pw += 4) {
            const float4 q = pad_slot ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(src + pw);
            float f[4] = {q.x, q.y, q.z, q.w};
            bf16x4 hv, lv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = f[e];
                if (p.normalize && !pad_slot) a = (a - mean) / stdv;
                bf16 hi2, lo2;
                split_bf16(a, hi2, lo2);
                hv[e] = hi2;
                lv[e] = lo2;
            }
            bf16* dst = p.out + a_pos<PLANES>(row, p.ld, kbase + pw);
            *reinterpret_cast<bf16x4*>(dst) = hv;
            if constexpr (PLANES == 2) *reinterpret_cast<bf16x4*>(dst + kLoOffset) = lv;
        }
        if (rem == 0) {  // zero the K padding (K = C*P*P rounded up to ld) once per row
            const int K = p.C * p.P * p.P;
            for (int k = K; k < p.ld; ++k) {
                bf16* z = p.out + a_pos<PLANES>(row, p.ld, k);
                *z = (bf16)0.f;
                if constexpr (PLANES == 2) z[kLoOffset] = (bf16)0.f;
            }
        }
    }
}

int launch_index_gather(const PatchGatherParams& p, const uint8_t* mask, int n_vis, int* perm, int* rank, int* err_rows, int planes, hipStream_t stream) {
    CWM_REQUIRE(p.P % 4 == 0 && p.W % 4 == 0, "index_gather: patch size and width must be multiples of 4");
    CWM_REQUIRE(p.C == 3 || !p.normalize, "index_gather: imagenet normalisation needs 3 channels");
    CWM_REQUIRE(p.ld >= p.C * p.P * p.P && p.ld % 4 == 0, "index_gather: bad ld");
    CWM_REQUIRE(mask && perm && err_rows && p.n_rows > 0 && p.n_rows <= 12000, "index_gather: bad argument (rows per sample: %d)", p.n_rows);
    // one gather item per thread.  (Capping the grid at one resident round of the chip -- every workgroup pays the mask scan once -- and letting the
    // grid-strided loop run twice measured SLOWER at batch 32: 28.6 against 22.2 us; the scan hides under the other workgroups' loads.)
    const int per_sample = (p.n_rows * p.C * p.P + 255) / 256;
    const dim3 grid((unsigned)per_sample, (unsigned)p.B);
    const size_t smem = (size_t)p.n_rows * sizeof(int);
    if (planes == 1)
        hipLaunchKernelGGL(index_gather_kernel<1>, grid, dim3(256), smem, stream, p, mask, n_vis, perm, rank, err_rows);
    else
        hipLaunchKernelGGL(index_gather_kernel<2>, grid, dim3(256), smem, stream, p, mask, n_vis, perm, rank, err_rows);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// decoder input, masked half: x_full[b][n_vis + j] = mask_token + pos[perm[b][n_vis + j]]
// (vmae.py:556-557).  The visible half is written by the encoder_to_decoder GEMM epilogue.
// ---------------------------------------------------------------------------------------------
// EIGHT rows per thread, all loads (perm -> positional row) issued
// before the first store.  One row per thread ran as ~4.6 rounds of short-lived waves, each a dependent perm -> pos -> store
// chain: 2.9 TB/s of stores (13.1 us for 38 MB, ViT-B/8 batch 32); with independent chains per thread the launch is one round.
__global__ __launch_bounds__(256) void fill_mask_tokens4_kernel(float* x_full, const float* mask_token, const float* pos, const int* perm, int Nt,
                                                                 int n_vis, int D, int64_t rows, int64_t rows_q) {
    const int d4 = D / 4;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rows_q * d4) return;
    const int64_t r0 = gid / d4;
    const int c4 = (int)(gid - r0 * d4);
    const int nm = Nt - n_vis;
    const float4 mt = *reinterpret_cast<const float4*>(mask_token + c4 * 4);
    constexpr int R = 8;  // rows per thread: the whole launch is resident at once (no second, nearly empty round of waves)
    size_t orow[R];
    int tau[R];
    bool live[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int64_t row = r0 + i * rows_q;
        live[i] = row < rows;
        const int64_t rr = live[i] ? row : 0;
        const int b = (int)(rr / nm), j = (int)(rr - (int64_t)b * nm);
        orow[i] = (size_t)b * Nt + n_vis + j;
        tau[i] = perm[orow[i]];
    }
    float4 pe[R];
#pragma unroll
    for (int i = 0; i < R; ++i) pe[i] = *reinterpret_cast<const float4*>(pos + (size_t)tau[i] * D + c4 * 4);
#pragma unroll
    for (int i = 0; i < R; ++i)
        if (live[i]) *reinterpret_cast<float4*>(x_full + orow[i] * D + c4 * 4) = make_float4(mt.x + pe[i].x, mt.y + pe[i].y, mt.z + pe[i].z, mt.w + pe[i].w);
}

int launch_fill_mask_tokens(float* x_full, const float* mask_token, const float* pos, const int* perm, int B, int Nt, int n_vis, int D, hipStream_t stream) {
    CWM_REQUIRE(D % 4 == 0, "fill_mask_tokens: D must be a multiple of 4");
    const int64_t rows = (int64_t)B * (Nt - n_vis), rows_q = (rows + 7) / 8;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(fill_mask_tokens4_kernel, dim3((unsigned)((rows_q * (D / 4) + 255) / 256)), dim3(256), 0, stream, x_full, mask_token, pos, perm, Nt,
                       n_vis, D, rows, rows_q);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Patch un-embed scatter.  Reference: `pred_patches_to_video` (prediction.py:245-259) with
// `Patchify` (patches.py:74,98-102): feature index f = (ph*P + pw)*C + c; visible patches take the
// RAW (un-normalised) wrapper input.  One thread per 4 horizontally adjacent output pixels.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unembed_kernel(const UnembedParams p) {
    const int w4 = p.W / 4;
    const int64_t total = (int64_t)p.B * p.T * p.C * p.H * w4;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    int64_t r = gid;
    const int x4 = (int)(r % w4); r /= w4;
    const int y = (int)(r % p.H); r /= p.H;
    const int c = (int)(r % p.C); r /= p.C;
    const int t = (int)(r % p.T);
    const int b = (int)(r / p.T);
    const int x0 = x4 * 4;
    const int gw = p.W / p.P;
    const int n = (p.H / p.P) * gw;
    const int Nt = p.T * n;
    const int tau = t * n + (y / p.P) * gw + (x0 / p.P);
    float4 o;
    if (p.mask[(size_t)b * Nt + tau]) {
        const int j = p.rank[(size_t)b * Nt + tau] - p.n_vis;
        const float* yp = p.y + ((size_t)b * p.Nm + j) * (p.P * p.P * p.C) + ((y % p.P) * p.P + (x0 % p.P)) * p.C + c;
        o = make_float4(yp[0], yp[p.C], yp[2 * p.C], yp[3 * p.C]);
    } else {
        o = *reinterpret_cast<const float4*>(p.x + b * p.sb + t * p.st + c * p.sc + (int64_t)y * p.W + x0);
    }
    *reinterpret_cast<float4*>(p.out + ((((size_t)b * p.T + t) * p.C + c) * p.H + y) * p.W + x0) = o;
}

// C == 3 (every predictor of the path): one thread per 4 horizontally adjacent pixels of ALL THREE channels.  A masked patch row is
// 4 px x 3 channels = 12 consecutive floats of the token (feature order (ph pw c), c fastest): three aligned 16-byte loads, transposed
// in registers, three 16-byte stores (one per channel plane).  The one-channel form above reads the same 48 bytes from three threads
// of three different waves with 4-byte loads at a 12-byte stride: 3.0 TB/s (25.7 us for 77 MB at ViT-B/8 batch 32).
__global__ __launch_bounds__(256) void unembed3_kernel(const UnembedParams p) {
    const int w4 = p.W / 4, hh = p.H / 2;  // two image rows (y, y + H/2) per thread: six independent 16-byte loads in flight
    const int64_t total = (int64_t)p.B * p.T * hh * w4;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    int64_t r = gid;
    const int x4 = (int)(r % w4); r /= w4;
    const int y0 = (int)(r % hh); r /= hh;
    const int t = (int)(r % p.T);
    const int b = (int)(r / p.T);
    const int x0 = x4 * 4;
    const int gw = p.W / p.P;
    const int n = (p.H / p.P) * gw;
    const int Nt = p.T * n;
    float4 o[2][3];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int y = y0 + k * hh;
        const int tau = t * n + (y / p.P) * gw + (x0 / p.P);
        if (p.mask[(size_t)b * Nt + tau]) {
            const int j = p.rank[(size_t)b * Nt + tau] - p.n_vis;
            const float4* yp = reinterpret_cast<const float4*>(p.y + ((size_t)b * p.Nm + j) * (p.P * p.P * 3) + ((y % p.P) * p.P + (x0 % p.P)) * 3);
            const float4 a = yp[0], bb = yp[1], c = yp[2];  // px0 (r g b) px1 (r | g b) px2 (r g | b) px3 (r g b)
            o[k][0] = make_float4(a.x, a.w, bb.z, c.y);
            o[k][1] = make_float4(a.y, bb.x, bb.w, c.z);
            o[k][2] = make_float4(a.z, bb.y, c.x, c.w);
        } else {
            const float* src = p.x + b * p.sb + t * p.st + (int64_t)y * p.W + x0;
#pragma unroll
            for (int c = 0; c < 3; ++c) o[k][c] = *reinterpret_cast<const float4*>(src + c * p.sc);
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float* dst = p.out + (((size_t)b * p.T + t) * 3 * p.H + y0 + k * hh) * p.W + x0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {  // (the predicted video is the call's result: written once, streaming stores)
            const f32x4 v = f32x4{o[k][c].x, o[k][c].y, o[k][c].z, o[k][c].w};
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst + (size_t)c * p.H * p.W));
        }
    }
}

int launch_unembed(const UnembedParams& p, hipStream_t stream) {
    CWM_REQUIRE(p.P % 4 == 0 && p.W % 4 == 0, "unembed: patch size and width must be multiples of 4");
    if (p.C == 3 && p.H % 2 == 0 && (((uintptr_t)p.y | (uintptr_t)p.out | (uintptr_t)p.x) & 15) == 0 && (p.sc % 4 == 0) && (p.sb % 4 == 0) && (p.st % 4 == 0)) {
        const int64_t total3 = (int64_t)p.B * p.T * (p.H / 2) * (p.W / 4);
        hipLaunchKernelGGL(unembed3_kernel, dim3((unsigned)((total3 + 255) / 256)), dim3(256), 0, stream, p);
        CWM_HIP_CHECK(hipGetLastError());
        return 0;
    }
    const int64_t total = (int64_t)p.B * p.T * p.C * p.H * (p.W / 4);
    hipLaunchKernelGGL(unembed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Motion-counterfactual prompt construction, vectorised over all B*S prompts.
// Reference: the per-sample Python loop of `create_motion_counterfactuals` (segmentation.py:324-338)
// calling `PatchPerturbation.forward` + `ShiftPatchesAndMask.perturb` (perturbation.py:99-113,
// 245-289) on a static movie (`make_static_movie`, prediction.py:731-739).  For prompt i = b*S+s:
//   frame `frame`: the destination patch (pi,pj) of every active patch (pi-dy, pj-dx) receives the
//   active patch's pixels; everything else is the (static) input.  Pure copies: bit-exact.
//   mask_out = (active ? masks : 1) & (shifted perturbation mask), destinations out of frame vanish.
// fix_passive: 0 = the movie as given; 1 = every frame := frame 0 (`make_static_movie`); 2 = `MakeStatic` (perturbation.py:120-145)
// applied before the shift: only the patches that `masks` leaves visible are replaced by their frame-0 pixels.
// ---------------------------------------------------------------------------------------------
// One thread = 4 horizontally adjacent pixels of FOUR consecutive image rows (y = 4 yq .. 4 yq + 3: one patch row block, P is a multiple of 4), so the
// source decision -- which patch, which frame -- is taken once for four independent 16-byte loads and stores.  (Until round 5: one 16-byte store per thread
// behind six integer divisions and a dependent mask byte load; profiles/r6_bench_prompt_build*.json.)  The prompts are written once and read once by the
// predictor's gather: streaming (non-temporal) stores.
__global__ __launch_bounds__(256) void shift_prompts_x_kernel(const ShiftPromptParams p) {
    const int w4 = p.W / 4, h4 = p.H / 4;
    const int64_t total = (int64_t)p.B * p.S * p.T * p.C * h4 * w4;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    int64_t r = gid;
    const int x4 = (int)(r % w4); r /= w4;
    const int yq = (int)(r % h4); r /= h4;
    const int c = (int)(r % p.C); r /= p.C;
    const int t = (int)(r % p.T);
    const int i = (int)(r / p.T);
    const int b = i / p.S;
    const int x0 = x4 * 4, y = yq * 4;
    int ft = p.fix_passive == 1 ? 0 : t;
    int sy = y, sx = x0;
    const int gw = p.W / p.P, gh = p.H / p.P, n = gh * gw;
    if (t == p.frame) {
        const int dy = p.shifts[2 * i], dx = p.shifts[2 * i + 1];
        const int pi = y / p.P - dy, pj = x0 / p.P - dx;
        if (pi >= 0 && pi < gh && pj >= 0 && pj < gw && p.active[(size_t)i * p.T * n + (size_t)p.frame * n + pi * gw + pj] == 0) {
            sy = y - dy * p.P;
            sx = x0 - dx * p.P;
        }
    }
    if (p.fix_passive == 2) {  // MakeStatic (perturbation.py:120-145): the patches `masks` leaves visible take their frame-0 pixels
        if (p.masks[(size_t)i * p.T * n + (size_t)t * n + (sy / p.P) * gw + sx / p.P] == 0) ft = 0;
    }
    const float* src = p.x + ((((size_t)b * p.T + ft) * p.C + c) * p.H + sy) * p.W + sx;
    float* dst = p.x_out + ((((size_t)i * p.T + t) * p.C + c) * p.H + y) * p.W + x0;
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4*>(src + (size_t)k * p.W);
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(v[k], reinterpret_cast<f32x4*>(dst + (size_t)k * p.W));
}

__global__ __launch_bounds__(256) void shift_prompts_mask_kernel(const ShiftPromptParams p) {
    const int gw = p.W / p.P, gh = p.H / p.P, n = gh * gw, Nt = p.T * n;
    const int64_t total = (int64_t)p.B * p.S * Nt;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int i = (int)(gid / Nt), tau = (int)(gid - (int64_t)i * Nt);
    const int t = tau / n, hw = tau - t * n;
    const uint8_t act = p.active[gid];
    const uint8_t m1 = act ? p.masks[gid] : 1;
    uint8_t mp = act;
    if (t == p.frame) {
        const int dy = p.shifts[2 * i], dx = p.shifts[2 * i + 1];
        const int pi = hw / gw - dy, pj = hw % gw - dx;
        mp = (pi >= 0 && pi < gh && pj >= 0 && pj < gw) ? p.active[(size_t)i * Nt + (size_t)t * n + pi * gw + pj] : 1;
    }
    p.mask_out[gid] = (m1 && mp) ? 1 : 0;
}

// The prompt table of the counterfactual batch (BASELINE configs[3]; interface.py:370-377: one active patch of frame `frame` per prompt, moved by (dy, dx)
// patches, nothing passive) -> the dense operands of the kernels above: passive[i] = "frame 0 visible, every later frame masked", active[i] = passive[i] with the
// prompt's patch cleared, shifts[i] = (dy, dx).  One launch instead of the ten tensor operations that built them on rank 0's critical path (dist.py prompt_hooks).
__global__ __launch_bounds__(256) void prompt_table_expand_kernel(const int* __restrict__ table, int S, int n, int gw, int T, int frame, uint8_t* __restrict__ active,
                                                                  uint8_t* __restrict__ passive, int* __restrict__ shifts) {
    const int Nt = T * n;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)S * Nt) return;
    const int i = (int)(gid / Nt), tau = (int)(gid - (int64_t)i * Nt);
    const int h = table[4 * i], w = table[4 * i + 1];
    const uint8_t pas = tau >= n ? 1 : 0;
    passive[gid] = pas;
    active[gid] = (tau == frame * n + h * gw + w) ? 0 : pas;
    if (tau < 2) shifts[2 * i + tau] = table[4 * i + 2 + tau];
}

int launch_prompt_table_expand(const int* table, int S, int n, int gw, int T, int frame, uint8_t* active, uint8_t* passive, int* shifts, hipStream_t stream) {
    const int64_t total = (int64_t)S * T * n;
    hipLaunchKernelGGL(prompt_table_expand_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, table, S, n, gw, T, frame, active, passive, shifts);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_shift_prompts(const ShiftPromptParams& p, hipStream_t stream) {
    CWM_REQUIRE(p.P % 4 == 0 && p.W % 4 == 0 && p.H % p.P == 0 && p.W % p.P == 0, "shift_prompts: bad patch/image size");
    CWM_REQUIRE(p.frame >= 0 && p.frame < p.T, "shift_prompts: frame out of range");
    CWM_REQUIRE(p.fix_passive >= 0 && p.fix_passive <= 2, "shift_prompts: fix_passive must be 0, 1 or 2");
    CWM_REQUIRE(p.H % 4 == 0, "shift_prompts: the image height must be a multiple of 4");
    const int64_t tx = (int64_t)p.B * p.S * p.T * p.C * (p.H / 4) * (p.W / 4);
    const int64_t tm = (int64_t)p.B * p.S * p.T * (p.H / p.P) * (p.W / p.P);
    // either output may be absent: the sharded loop builds the masks of ALL prompts on rank 0 (the rectangulariser is the one cross-row step) but frames only for
    // the rows a rank predicts (dist.py)
    if (p.x_out) hipLaunchKernelGGL(shift_prompts_x_kernel, dim3((unsigned)((tx + 255) / 256)), dim3(256), 0, stream, p);
    if (p.mask_out) hipLaunchKernelGGL(shift_prompts_mask_kernel, dim3((unsigned)((tm + 255) / 256)), dim3(256), 0, stream, p);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// `RectangularizeMasks` on device masks (masking.py:90-132) without reading the masks back: the host needs only the per-row COUNTS to decide
// the target and to draw the reference's `torch.randperm(#candidates)[:surplus]` per changed row (the draws depend on the counts alone); the
// picks -- "the k-th masked (or visible) token of row r in ascending order" -- come back as a small table and are applied here, every pick of a
// row against the row as it was BEFORE any of them (the reference indexes one `torch.where` list per row).  Integer-exact.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_row_counts_kernel(const uint8_t* __restrict__ mask, int Nt, int* __restrict__ counts) {
    __shared__ int red[4];
    const uint8_t* m = mask + (size_t)blockIdx.x * Nt;
    int c = 0;
    for (int i = threadIdx.x; i < Nt; i += 256) c += (m[i] != 0);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// table: [R | rows[R] | offsets[R + 1] | to_value[R] | picks[offsets[R]]]; one workgroup per changed row
constexpr int kFlipMaxTokens = 16384;
__global__ __launch_bounds__(256) void mask_flip_picks_kernel(uint8_t* __restrict__ mask, int Nt, const int* __restrict__ table) {
    __shared__ unsigned bits[kFlipMaxTokens / 32];
    __shared__ int wave_tot[4];
    const int R = table[0], r = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int row = table[1 + r], p0 = table[1 + R + r], p1 = table[1 + R + r + 1], to = table[2 + 2 * R + r];
    const int* picks = table + 2 + 3 * R;
    uint8_t* m = mask + (size_t)row * Nt;
    for (int i = t; i < (Nt + 31) / 32; i += 256) bits[i] = 0u;
    __syncthreads();
    for (int q = p0 + t; q < p1; q += 256) {
        const int k = picks[q];
        if (k >= 0 && k < Nt) atomicOr(&bits[k >> 5], 1u << (k & 31));
    }
    const int per = (Nt + 255) / 256;
    const int lo = min(t * per, Nt), hi = min(lo + per, Nt);
    int c = 0;
    for (int i = lo; i < hi; ++i) c += ((m[i] != 0) != (to != 0));  // candidates: tokens whose state is not `to` yet
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();  // (also: the bitmap is complete)
    int v = incl - c;
    for (int w2 = 0; w2 < wave; ++w2) v += wave_tot[w2];
    for (int i = lo; i < hi; ++i) {
        if ((m[i] != 0) != (to != 0)) {
            if (bits[v >> 5] & (1u << (v & 31))) m[i] = (uint8_t)(to ? 1 : 0);
            ++v;
        }
    }
}

int launch_mask_row_counts(const uint8_t* mask, int B, int Nt, int* counts, hipStream_t stream) {
    hipLaunchKernelGGL(mask_row_counts_kernel, dim3((unsigned)B), dim3(256), 0, stream, mask, Nt, counts);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_mask_flip_picks(uint8_t* mask, int Nt, const int* table, int n_rows, hipStream_t stream) {
    CWM_REQUIRE(Nt <= kFlipMaxTokens, "mask_flip_picks: rows of %d tokens exceed the %d-token pick bitmap", Nt, kFlipMaxTokens);
    if (n_rows == 0) return 0;
    hipLaunchKernelGGL(mask_flip_picks_kernel, dim3((unsigned)n_rows), dim3(256), 0, stream, mask, Nt, table);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
__global__ void split_bf16_kernel(const float* x, int64_t n, bf16* hi, bf16* lo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bf16 h, l;
    split_bf16(x[i], h, l);
    hi[i] = h;
    if (lo) lo[i] = l;
}

int launch_split_bf16(const float* x, int64_t n, bf16* hi, bf16* lo, hipStream_t stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, n, hi, lo);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace cwm
