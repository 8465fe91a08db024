// MFMA attention kernels of the IMU-conditioned conjoined predictor (BASELINE configs[4]) for CDNA4 (gfx950):
//   cross_attn_mfma_kernel   `BidirectionalCrossAttention.forward` (transformer.py:314-378, shared_similarity=False), both directions
//   cross_attn_combine_kernel merges the per-wave partials of the context-side update
//   small_attn_mfma_kernel   `Attention.forward` (VideoMAE/utils.py:87-121) of the context stream (<= 64 tokens, head_dim 32)
//
// Cross attention.  Per head h the 2*hd-wide slice of qk / qk_src splits into
//   [0,hd):   attn   = softmax_M( scale * qk1 . qk_src1^T )   -> y     = attn   . v_src   (main update,    N x M scores)
//   [hd,2hd): attn_s = softmax_N( scale * qk_src2 . qk2^T )   -> y_src = attn_s . v       (context update, M x N scores)
// with N = 3140 .. 6336 main tokens against M = 25 / 50 context tokens.  The arithmetic is nothing (8 GFLOP per call); what a call has
// to do is stream the main-stream projections once: qk 2D + v D columns in, y D columns out = 16 bytes x D per token (650 MB per
// encoder call at batch 16 = 130 us at 5 TB/s).  The first form of these kernels (conj_kernels.hip: fp32 VALU dot products against LDS
// broadcasts, the N x M src-side scores staged through HBM) ran at 1.1 ms per call, 10 % of the model's step.  Here both directions are
// the flash-attention dataflow of attention.hip on 32x32x16 MFMAs, one 32-token chunk per wave and step:
//   role A (main update)     queries = the chunk's tokens, keys = the context (ONE key tile: 32 or 64 padded context tokens)
//        S^T[ctx][tok] = K1 . Q1^T         A = context fragments (LDS), B = token fragments STRAIGHT FROM HBM (the projections are
//                                          written in the GEMM A-operand layout, common.h a_pos: a lane's 8 k values are one 16-byte load)
//        softmax over ctx is lane-local (+ one lane^32 exchange); O^T[d][tok] = V_src^T . P^T with the S^T accumulator as B operand
//   role B (context update)  queries = the context, keys = the chunk's tokens (flash accumulation over the wave's chunks)
//        S^T[tok][ctx] = Q2 . K2^T         A = token fragments from HBM, B = context fragments (LDS)
//        online softmax over tokens per context column; O^T[d][ctx] += V^T . P^T, V^T by the hardware transpose read
//        (ds_read_b64_tr_b16) from a wave-private LDS image of the chunk's V rows, 64 d at a time
//   every (batch, head) is cut into kCrossSplit wave-sized shares of its chunks; a wave of role B leaves (max, sum, O^T) of its share,
//   cross_attn_combine_kernel merges the shares (the role A half of the grid has nothing to merge).
// The two roles read disjoint columns of qk (q1 | q2), so running them as two halves of one grid costs no extra traffic.
// PLANES == 2 is the split-bf16 "parity" arithmetic (hi*hi + hi*lo + lo*hi) of every other MFMA product on the path.
#include <math.h>

#include "attention_device.h"

namespace cwm {

constexpr int kCrossSplit2 = 16;  // wave-sized shares per (batch, head): grid.x = kCrossSplit2 / 4 workgroups of 4 waves

namespace {

// position of context token m in the k order the accumulator-as-operand trick asks for: within a block of 32, the fragment of
// k-step s holds, for lane half hh, keys 16 s + 8 (j >> 2) + 4 hh + (j & 3) at element j  ->  stored at 16 s + 8 hh + j
__device__ __forceinline__ int ctx_pos(int m) {
    const int w = m & 31, r = w & 15;
    return (m & ~31) + (w & 16) + 8 * ((r >> 2) & 1) + ((r >> 3) << 2) + (r & 3);
}

template <int PLANES>
__device__ __forceinline__ void p_fragments(const f32x16& s, int ks, bf16x8& ph, bf16x8& plo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float pv = s[8 * ks + j];
        const bf16 hi = (bf16)pv;
        ph[j] = hi;
        if constexpr (PLANES == 2) plo[j] = (bf16)(pv - (float)hi);
    }
}

__device__ __forceinline__ float sum_lane_xor32(float x) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}

}  // namespace

// partial[((bh * kCrossSplit2 + split) * M + m) * (hd + 4) + {d | hd: local max (log2 domain, scaled) | hd + 1: local sum}]
template <int PLANES, int NDB, int MT, int ROLE>
__global__ __launch_bounds__(ROLE == 0 ? 512 : 256, 2) void cross_attn_mfma_kernel(const CrossAttnParams p) {
    constexpr int NW = ROLE == 0 ? 8 : 4;                            // role A needs 78 registers: 8 waves share the context's LDS image
    constexpr int NSPLIT = ROLE == 0 ? 2 * kCrossSplit2 : kCrossSplit2;  // wave-sized shares of a (batch, head)'s chunks
    constexpr int HD = NDB * 32, KS = HD / 16;                       // head_dim, 16-wide k-steps of a score product
    constexpr int KG = (KS % 4 == 0) ? 4 : (KS % 3 == 0 ? 3 : KS);   // k-steps whose token fragments are requested together (role A)
    constexpr int NG = KS / KG;
    constexpr int KGB = 2, NGB = KS / KGB;                           // role B keeps 96 accumulator registers across chunks: smaller groups
    constexpr int KROW = HD * 2 + 16;        // context K rows in LDS: + 16 bytes, so that the 16 rows of a ds_read_b128 lane group hit 16 bank quads
    constexpr int VROW = 64 * MT + 16;       // V_src^T rows ([d][context position])
    constexpr int KPLANE = 32 * MT * KROW, VPLANE = HD * VROW;
    constexpr int VT = 32 * 128;             // role B: one plane of a wave's V image (32 tokens x 64 d)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qcol = lane & 31, hh = lane >> 5;
    constexpr int role = ROLE;  // 0: main-stream update (role A), 1: context update (role B); one launch each, back to back
    const int bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads;
    const int D = p.heads * HD, N = p.N, M = p.M;
    const int split = blockIdx.x * NW + wave;
    const int NC = (N + 31) / 32;
    const float c2 = p.scale * 1.4426950408889634f;  // softmax in the exp2 domain

    // ---- context operands of this (batch, head) into LDS: K1 (role A) / K2 (role B) rows, V_src^T (role A) ----
    char* kl = smem;
    {
        const float* src = p.qk_src + (size_t)b * M * 2 * D + h * 2 * HD + role * HD;
        for (int i = tid; i < 32 * MT * HD; i += NW * 64) {
            const int m = i / HD, d = i - m * HD;
            const float v = m < M ? src[(size_t)m * 2 * D + d] : 0.f;
            bf16 hi, lo;
            split_bf16(v, hi, lo);
            *reinterpret_cast<bf16*>(kl + m * KROW + d * 2) = hi;
            if constexpr (PLANES == 2) *reinterpret_cast<bf16*>(kl + KPLANE + m * KROW + d * 2) = lo;
        }
    }
    char* vl = smem + PLANES * KPLANE;  // role A: V_src^T planes; role B: the waves' V images
    if constexpr (role == 0) {
        const float* src = p.v_src + (size_t)b * M * D + h * HD;
        for (int i = tid; i < 32 * MT * HD; i += NW * 64) {
            const int m = i / HD, d = i - m * HD;
            const float v = m < M ? src[(size_t)m * D + d] : 0.f;
            bf16 hi, lo;
            split_bf16(v, hi, lo);
            const int off = d * VROW + ctx_pos(m) * 2;
            *reinterpret_cast<bf16*>(vl + off) = hi;
            if constexpr (PLANES == 2) *reinterpret_cast<bf16*>(vl + VPLANE + off) = lo;
        }
    }
    __syncthreads();

    // token fragments of k-step ks (hi [, lo]): 8 consecutive k of token row `row`, first column `col0` of the head's q1 / q2 slice
    auto load_q = [&](auto& qf, int64_t row, int col0, int g) {
        constexpr int G = sizeof(qf) / sizeof(qf[0]);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const bf16* src = p.qk_op + a_pos<PLANES>(row, 2 * D, col0 + (g * G + i) * 16 + hh * 8);
            qf[i][0] = *reinterpret_cast<const bf16x8*>(src);
            if constexpr (PLANES == 2) qf[i][1] = *reinterpret_cast<const bf16x8*>(src + kLoOffset);
        }
    };

    if constexpr (role == 0) {
        // =================================== role A: main-stream update ===================================
        for (int c = split; c < NC; c += NSPLIT) {
            const int tok = c * 32 + qcol;
            const int64_t row = (int64_t)b * N + min(tok, N - 1);
            f32x16 sacc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[mt][r] = 0.f;
            bf16x8 qf[2][KG][PLANES];
            load_q(qf[0], row, h * 2 * HD, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) load_q(qf[(g + 1) & 1], row, h * 2 * HD, g + 1);
#pragma unroll
                for (int i = 0; i < KG; ++i)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const char* ka = kl + (mt * 32 + qcol) * KROW + ((g * KG + i) * 16 + hh * 8) * 2;
                        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(ka);
                        if constexpr (PLANES == 2) {
                            const bf16x8 klo = *reinterpret_cast<const bf16x8*>(ka + KPLANE);
                            sacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(klo, qf[g & 1][i][0], sacc[mt], 0, 0, 0);
                            sacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[g & 1][i][PLANES - 1], sacc[mt], 0, 0, 0);
                        }
                        sacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[g & 1][i][0], sacc[mt], 0, 0, 0);
                    }
            }
            // softmax over the context (rows of S^T): 16 rows per block in this lane, the other 16 in lane ^ 32
            float mx = -INFINITY;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ctx = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (ctx >= M) sacc[mt][r] = -INFINITY;
                    mx = fmaxf(mx, sacc[mt][r]);
                }
            mx = max_lane_xor32(mx);
            const float mc = mx * c2;
            float rowsum = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[mt][r], c2, -mc));
                    sacc[mt][r] = pv;
                    rowsum += pv;
                }
            const float inv = 1.0f / sum_lane_xor32(rowsum);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[mt][r] *= inv;
            // O^T[d][tok] = V_src^T . P^T, two d-blocks (two independent accumulator chains) at a time, written out as they finish
            bf16x8 ph[2 * MT], plo[2 * MT];
#pragma unroll
            for (int ks = 0; ks < 2 * MT; ++ks) p_fragments<PLANES>(sacc[ks >> 1], ks & 1, ph[ks], plo[ks]);
#pragma unroll
            for (int db0 = 0; db0 < NDB; db0 += 2) {
                f32x16 oacc[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 2 * MT; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (db0 + i >= NDB) continue;
                        const char* va = vl + ((db0 + i) * 32 + qcol) * VROW + (ks * 16 + hh * 8) * 2;
                        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(va);
                        if constexpr (PLANES == 2) {
                            const bf16x8 vlo = *reinterpret_cast<const bf16x8*>(va + VPLANE);
                            oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vlo, ph[ks], oacc[i], 0, 0, 0);
                            oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, plo[ks], oacc[i], 0, 0, 0);
                        }
                        oacc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, ph[ks], oacc[i], 0, 0, 0);
                    }
                if (tok < N) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (db0 + i >= NDB) continue;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            bf16x4 hi4, lo4;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float v = oacc[i][4 * g + e];
                                const bf16 hi = (bf16)v;
                                hi4[e] = hi;
                                if constexpr (PLANES == 2) lo4[e] = (bf16)(v - (float)hi);
                            }
                            bf16* dst = p.y + a_pos<PLANES>(row, D, h * HD + (db0 + i) * 32 + 8 * g + 4 * hh);  // GEMM A-operand layout (the projection reads it)
                            *reinterpret_cast<bf16x4*>(dst) = hi4;
                            if constexpr (PLANES == 2) *reinterpret_cast<bf16x4*>(dst + kLoOffset) = lo4;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
    // =================================== role B: context update ===================================
    char* vimg = vl + wave * (PLANES * VT);
    int v_base[2];
    {
        const int g = lane >> 4, q = (lane >> 2) & 3, pc = lane & 3;  // see attention.hip: the transposed read of a [key][64 d] image
#pragma unroll
        for (int db = 0; db < 2; ++db) v_base[db] = lds_off_v(4 * (g >> 1) + q, db * 4 + (g & 1) * 2 + (pc >> 1)) + (pc & 1) * 8;
    }
    f32x16 oacc[MT][NDB];
    float m_run[MT], l_run[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        m_run[mt] = -1e30f;
        l_run[mt] = 0.f;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[mt][db][r] = 0.f;
    }
    for (int c = split; c < NC; c += kCrossSplit2) {
        const int tok0 = c * 32;
        const int64_t row = (int64_t)b * N + min(tok0 + qcol, N - 1);
        f32x16 sacc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[mt][r] = 0.f;
        bf16x8 qf[2][KGB][PLANES];
        load_q(qf[0], row, h * 2 * HD + HD, 0);
#pragma unroll
        for (int g = 0; g < NGB; ++g) {
            if (g + 1 < NGB) load_q(qf[(g + 1) & 1], row, h * 2 * HD + HD, g + 1);
#pragma unroll
            for (int i = 0; i < KGB; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const char* ka = kl + (mt * 32 + qcol) * KROW + ((g * KGB + i) * 16 + hh * 8) * 2;
                    const bf16x8 kh = *reinterpret_cast<const bf16x8*>(ka);
                    if constexpr (PLANES == 2) {
                        const bf16x8 klo = *reinterpret_cast<const bf16x8*>(ka + KPLANE);
                        sacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[g & 1][i][PLANES - 1], kh, sacc[mt], 0, 0, 0);
                        sacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[g & 1][i][0], klo, sacc[mt], 0, 0, 0);
                    }
                    sacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[g & 1][i][0], kh, sacc[mt], 0, 0, 0);
                }
        }
        // online softmax over the tokens (rows of S^T) per context column
        const bool ragged = tok0 + 32 > N;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (ragged && tok0 + (r & 3) + 8 * (r >> 2) + 4 * hh >= N) sacc[mt][r] = -INFINITY;
                mx = fmaxf(mx, sacc[mt][r]);
            }
            mx = max_lane_xor32(mx);
            const float m_new = fmaxf(m_run[mt], mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run[mt] - m_new) * c2);
            m_run[mt] = m_new;
            const float mc = m_new * c2;
            float rowsum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[mt][r], c2, -mc));
                sacc[mt][r] = pv;
                rowsum += pv;
            }
            l_run[mt] = l_run[mt] * alpha + rowsum;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[mt][db][r] *= alpha;
        }
        bf16x8 ph[MT][2], plo[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) p_fragments<PLANES>(sacc[mt], ks, ph[mt][ks], plo[mt][ks]);
        // O^T[d][ctx] += V^T . P^T, 64 d of the chunk's V rows at a time through the wave's LDS image
#pragma unroll
        for (int t = 0; t < (NDB + 1) / 2; ++t) {
            constexpr int NCH = 8;                           // 16-byte chunks per image row
            const bool half_tile = 2 * t + 1 >= NDB;         // the last image of an odd NDB holds 32 d only
            u32x4 rv[PLANES][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = lane + 64 * i, r = idx / NCH, ch = idx % NCH;
                if (!(half_tile && ch >= 4)) {
                    const bf16* src = p.v_op + a_pos<PLANES>((int64_t)b * N + min(tok0 + r, N - 1), D, h * HD + t * 64 + ch * 8);
                    rv[0][i] = *reinterpret_cast<const u32x4*>(src);
                    if constexpr (PLANES == 2) rv[1][i] = *reinterpret_cast<const u32x4*>(src + kLoOffset);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = lane + 64 * i, r = idx / NCH, ch = idx % NCH;
                if (!(half_tile && ch >= 4)) {
#pragma unroll
                    for (int pl = 0; pl < PLANES; ++pl) *reinterpret_cast<u32x4*>(vimg + pl * VT + lds_off_v(r, ch)) = rv[pl][i];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the image is wave-private: program order + the in-order LDS queue
#pragma unroll
            for (int db2 = 0; db2 < 2; ++db2) {
                if (2 * t + db2 >= NDB) continue;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x4 v0[PLANES], v1[PLANES];
#pragma unroll
                    for (int pl = 0; pl < PLANES; ++pl) {
                        const char* vb = vimg + pl * VT + v_base[db2] + ks * 2048;
                        v0[pl] = lds_read_tr16(vb);
                        v1[pl] = lds_read_tr16(vb + 1024);
                    }
                    const bf16x8 vf = __builtin_shufflevector(v0[0], v1[0], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        f32x16& o = oacc[mt][2 * t + db2];
                        if constexpr (PLANES == 2) {
                            const bf16x8 vlo = __builtin_shufflevector(v0[PLANES - 1], v1[PLANES - 1], 0, 1, 2, 3, 4, 5, 6, 7);
                            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vlo, ph[mt][ks], o, 0, 0, 0);
                            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, plo[mt][ks], o, 0, 0, 0);
                        }
                        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, ph[mt][ks], o, 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_sched_barrier(0);  // (keeps the next image's loads from being hoisted above this one's MFMAs: registers)
        }
    }
    // the wave's share: (max, sum, O^T) per context token
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int ctx = mt * 32 + qcol;
        const float l_tot = sum_lane_xor32(l_run[mt]);
        if (ctx < M) {
            float* out = p.partial + ((size_t)(bh * kCrossSplit2 + split) * M + ctx) * (HD + 4);
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(out + db * 32 + 8 * g + 4 * hh) = f32x4{oacc[mt][db][4 * g], oacc[mt][db][4 * g + 1], oacc[mt][db][4 * g + 2], oacc[mt][db][4 * g + 3]};
            if (hh == 0) {
                out[HD] = m_run[mt] * c2;
                out[HD + 1] = l_tot;
            }
        }
    }
    }  // role B
}

template <int PLANES>
__global__ __launch_bounds__(256) void cross_attn_combine_kernel(const CrossAttnParams p) {
    const int hd = p.head_dim, D = p.heads * hd, M = p.M;
    const int m = blockIdx.x, bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads, t = threadIdx.x;
    if (t >= hd) return;
    const float* part = p.partial + ((size_t)bh * kCrossSplit2 * M + m) * (hd + 4);
    const size_t sstride = (size_t)M * (hd + 4);
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < kCrossSplit2; ++s) mx = fmaxf(mx, part[s * sstride + hd]);
    float l = 0.f, acc = 0.f;
#pragma unroll
    for (int s = 0; s < kCrossSplit2; ++s) {
        const float w = __builtin_amdgcn_exp2f(part[s * sstride + hd] - mx);  // (a share without chunks has max -1e30 c: weight 0)
        l = fmaf(part[s * sstride + hd + 1], w, l);
        acc = fmaf(part[s * sstride + t], w, acc);
    }
    bf16 hi, lo;
    split_bf16(acc / l, hi, lo);
    bf16* dst = p.y_src + a_pos<PLANES>((int64_t)b * M + m, D, h * hd + t);
    *dst = hi;
    if constexpr (PLANES == 2) dst[kLoOffset] = lo;
}

// ---------------------------------------------------------------------------------------------
// Short-sequence self-attention of the context stream (25 / 50 tokens, head_dim 32): ONE wave per (batch, head), everything in
// registers.  S^T = K . Q^T and O^T = V^T . P^T on 32x32x16 MFMAs exactly as attention.hip, the fragments converted from the fp32
// qkv rows on the way in (q scaled by hd^-0.5 in fp32 first, VideoMAE/utils.py:94-113).  The VALU form (conj_kernels.hip: one thread
// per query, serial over keys against LDS broadcasts) took 135 us per call for 4 MFLOP.
// ---------------------------------------------------------------------------------------------
template <int PLANES, int NT>  // NT = 32-token blocks (1 or 2)
__global__ __launch_bounds__(64) void small_attn_mfma_kernel(const SmallAttnParams p) {
    constexpr int HD = 32;
    const int lane = threadIdx.x, qcol = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.x, b = bh / p.heads, h = bh - b * p.heads;
    const int N = p.n_tok, D = p.heads * HD;
    const float* base = p.qkv + (size_t)b * N * 3 * D + h * HD;
    const float scale = 1.0f / sqrtf((float)HD);
    auto frag = [&](const float* src, float mul, bf16x8& hi, bf16x8& lo) {  // 8 consecutive fp32 -> bf16 hi [, lo]
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), c = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = (e < 4 ? a[e] : c[e - 4]) * mul;
            const bf16 x = (bf16)v;
            hi[e] = x;
            if constexpr (PLANES == 2) lo[e] = (bf16)(v - (float)x);
        }
    };
    bf16x8 qh[NT][2], ql[NT][2], kh[NT][2], klo[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float* r = base + (size_t)min(t * 32 + qcol, N - 1) * 3 * D + ks * 16 + hh * 8;
            frag(r, scale, qh[t][ks], ql[t][ks]);
            frag(r + D, 1.0f, kh[t][ks], klo[t][ks]);
        }
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
        if (qt * 32 >= N) break;
        f32x16 sacc[NT];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if constexpr (PLANES == 2) {
                    sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(klo[kt][ks], qh[qt][ks], sacc[kt], 0, 0, 0);
                    sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[kt][ks], ql[qt][ks], sacc[kt], 0, 0, 0);
                }
                sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh[kt][ks], qh[qt][ks], sacc[kt], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= N) sacc[kt][r] = -INFINITY;
                mx = fmaxf(mx, sacc[kt][r]);
            }
        }
        mx = max_lane_xor32(mx);
        float rowsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f((sacc[kt][r] - mx) * 1.4426950408889634f);
                sacc[kt][r] = pv;
                rowsum += pv;
            }
        const float inv = 1.0f / sum_lane_xor32(rowsum);
        f32x16 oacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 ph, plo, vh, vlo;
                p_fragments<PLANES>(sacc[kt], ks, ph, plo);
#pragma unroll
                for (int j = 0; j < 8; ++j) {  // V^T fragment: lane (d = qcol, half hh), element j = key 16 ks + 8 (j >> 2) + 4 hh + (j & 3)
                    const int key = kt * 32 + 16 * ks + 8 * (j >> 2) + 4 * hh + (j & 3);
                    const float v = base[(size_t)min(key, N - 1) * 3 * D + 2 * D + qcol];
                    const bf16 x = (bf16)v;
                    vh[j] = x;
                    if constexpr (PLANES == 2) vlo[j] = (bf16)(v - (float)x);
                }
                if constexpr (PLANES == 2) {
                    oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vlo, ph, oacc, 0, 0, 0);
                    oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, plo, oacc, 0, 0, 0);
                }
                oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, oacc, 0, 0, 0);
            }
        const int q = qt * 32 + qcol;
        if (q < N) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 hi4, lo4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = oacc[4 * g + e] * inv;
                    const bf16 x = (bf16)v;
                    hi4[e] = x;
                    if constexpr (PLANES == 2) lo4[e] = (bf16)(v - (float)x);
                }
                bf16* dst = p.o + a_pos<PLANES>((int64_t)b * N + q, p.ldo, h * HD + 8 * g + 4 * hh);
                *reinterpret_cast<bf16x4*>(dst) = hi4;
                if constexpr (PLANES == 2) *reinterpret_cast<bf16x4*>(dst + kLoOffset) = lo4;
            }
        }
    }
}

bool small_attention_mfma_ok(int n_tok, int head_dim) { return head_dim == 32 && n_tok >= 1 && n_tok <= 64; }

int launch_small_attention_mfma(const SmallAttnParams& p, int planes, hipStream_t stream) {
    CWM_REQUIRE(small_attention_mfma_ok(p.n_tok, p.head_dim), "small_attention (MFMA): needs head_dim 32 and n_tok <= 64 (got %d, %d)", p.head_dim, p.n_tok);
    CWM_REQUIRE((p.heads * p.head_dim) % 4 == 0 && p.ldo % 32 == 0, "small_attention (MFMA): bad widths");
    const dim3 grid(p.B * p.heads), block(64);
    if (planes == 2) {
        if (p.n_tok <= 32) hipLaunchKernelGGL((small_attn_mfma_kernel<2, 1>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((small_attn_mfma_kernel<2, 2>), grid, block, 0, stream, p);
    } else {
        if (p.n_tok <= 32) hipLaunchKernelGGL((small_attn_mfma_kernel<1, 1>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((small_attn_mfma_kernel<1, 2>), grid, block, 0, stream, p);
    }
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

// (head_dim 192 against more than 32 context tokens would need more than 256 registers in role B: that shape stays on the VALU kernels)
bool cross_attention_mfma_ok(int head_dim, int M) { return (head_dim == 32 || head_dim == 96 || head_dim == 192) && M >= 1 && M <= (head_dim == 192 ? 32 : 64); }
// ... and the kernel's 32-bit element offsets must hold the lane's projections (B rows of N tokens, 4 * heads * head_dim elements each):
// beyond it (about 100 samples per lane at N = 6336, D = 768) the caller falls back to the VALU kernels instead of failing the forward
bool cross_attention_mfma_fits(int B, int N, int heads, int head_dim) { return N >= 1 && (int64_t)B * N * 4 * heads * head_dim < (1ll << 31); }
size_t cross_attention_mfma_partial_floats(int B, int heads, int M, int head_dim) { return (size_t)B * heads * kCrossSplit2 * M * (head_dim + 4); }

template <int PLANES, int NDB, int MT>
static int launch_cross_mfma_t(const CrossAttnParams& p, hipStream_t stream, hipStream_t stream_b, int roles) {
    constexpr int HD = NDB * 32, KPLANE = 32 * MT * (HD * 2 + 16), VPLANE = HD * (64 * MT + 16);
    constexpr int smem_a = PLANES * (KPLANE + VPLANE), smem_b = PLANES * KPLANE + 4 * PLANES * 4096;
    auto ka = cross_attn_mfma_kernel<PLANES, NDB, MT, 0>;
    auto kb = cross_attn_mfma_kernel<PLANES, NDB, MT, 1>;
    if (int rc = cwm_set_max_lds((const void*)ka, smem_a)) return rc;
    if (int rc = cwm_set_max_lds((const void*)kb, smem_b)) return rc;
    // role A (main-stream update) on `stream`; role B (context update) + its combine on `stream_b` (may be the same stream)
    if (roles & 1) hipLaunchKernelGGL(ka, dim3(2 * kCrossSplit2 / 8, p.B * p.heads), dim3(512), smem_a, stream, p);
    if (roles & 2) {
        hipLaunchKernelGGL(kb, dim3(kCrossSplit2 / 4, p.B * p.heads), dim3(256), smem_b, stream_b, p);
        hipLaunchKernelGGL(cross_attn_combine_kernel<PLANES>, dim3(p.M, p.B * p.heads), dim3(256), 0, stream_b, p);
    }
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_cross_attention_mfma(const CrossAttnParams& p, int planes, hipStream_t stream) { return launch_cross_attention_mfma_roles(p, planes, stream, stream, 3); }

// roles: bit 0 = role A (main-stream update, on stream_a), bit 1 = role B + combine (context update, on stream_b).  The two roles touch
// disjoint outputs and may run concurrently (conj_model.hip puts role B on the context stream's side stream).
int launch_cross_attention_mfma_roles(const CrossAttnParams& p, int planes, hipStream_t stream_a, hipStream_t stream_b, int roles) {
    // (a null stream handle is a valid stream: the role mask, not the handle, says what to launch)
    CWM_REQUIRE(cross_attention_mfma_ok(p.head_dim, p.M), "cross_attention (MFMA): head_dim %d / M %d not supported", p.head_dim, p.M);
    CWM_REQUIRE(p.qk_op && p.v_op && p.qk_src && p.v_src && p.y && p.y_src && p.partial, "cross_attention (MFMA): null argument");
    CWM_REQUIRE(cross_attention_mfma_fits(p.B, p.N, p.heads, p.head_dim), "cross_attention (MFMA): problem too large for 32-bit offsets");
    const int ndb = p.head_dim / 32, mt = p.M > 32 ? 2 : 1;
#define CWM_CROSS_CASE(PL, NDB, MT) \
    if (planes == PL && ndb == NDB && mt == MT) return launch_cross_mfma_t<PL, NDB, MT>(p, stream_a, stream_b, roles);
    CWM_CROSS_CASE(2, 6, 1) CWM_CROSS_CASE(2, 3, 1) CWM_CROSS_CASE(2, 3, 2) CWM_CROSS_CASE(2, 1, 1) CWM_CROSS_CASE(2, 1, 2)
    CWM_CROSS_CASE(1, 6, 1) CWM_CROSS_CASE(1, 3, 1) CWM_CROSS_CASE(1, 3, 2) CWM_CROSS_CASE(1, 1, 1) CWM_CROSS_CASE(1, 1, 2)
#undef CWM_CROSS_CASE
    cwm_set_error("cross_attention (MFMA): no kernel for planes %d, head_dim %d, M %d", planes, p.head_dim, p.M);
    return -1;
}

}  // namespace cwm
