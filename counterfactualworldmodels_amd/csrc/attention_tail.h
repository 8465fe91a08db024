// The ragged last query tile of a sequence whose length is not a multiple of 128 (ViT-B/8: N = 792 = 6 x 128 + 24, decoder 1568 =
// 12 x 128 + 32), when it holds at most 32 query rows: ONE 32-row query block, i.e. work for one of the workgroup's four waves.
//
// In the regular schedule that block keeps one wave busy for a whole key loop while three waves only help to stage tiles, and
// because the ragged tiles are dispatched last (attention_device.h) the launch ends with a phase in which every resident workgroup
// runs a single wave.  Measured gain of the split (profiles/r3d_*attn_tail*.log): 0.4 % on the batch-32 encoder launch, 3 % on the
// half-batch launches of a two-lane call, +0.4 % on the step -- far less than the slot arithmetic suggests (a wave that is alone on
// its SIMD runs its key loop faster, so the single-wave phase was shorter than a full workgroup period).  Here the four
// waves of the workgroup SPLIT THE KEYS of the same 32 queries: a pass stages two key tiles (the two LDS stages of the regular
// kernel hold one tile each), wave w takes the 32-key half (w & 1) of tile 2 pass + (w >> 1)
// through the per-wave arithmetic of attention_kernel -- S^T = K Q^T on 32x32x16 MFMAs, lane-local online softmax, O^T += V^T P^T
// with the hardware transpose read --, and the four (max, sum, O^T) partials are merged pairwise through LDS at the end.
// The result differs from the unsplit schedule by the order of the fp32 additions and by WHERE the split-bf16 rounding of P falls (every
// wave exponentiates against its own running maximum): ~1e-5 relative, inside the parity tolerance like the regular schedule;
// option "attn_tail" = 0 (cwm_model_set_option) keeps the regular schedule (the kernels' bitwise cross-check uses it).
#pragma once
#include "attention_device.h"

namespace cwm {

template <int PLANES>
__device__ __forceinline__ void attention_tail_block(const AttnParams& p, char* smem, int bh, int q0, int NQ) {
    constexpr int TILE_BYTES = 64 * 64 * 2;               // one 64x64 bf16 tile
    constexpr int STAGE_BYTES = TILE_BYTES * 2 * PLANES;  // K planes, then V planes of one key tile
    constexpr float kLog2eT = 1.4426950408889634f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qcol = lane & 31, hh = lane >> 5;
    const int N = p.n_tok;
    const int b = bh / p.heads, h = bh - b * p.heads;
    const int kb = wave & 1, slot = wave >> 1;  // this wave's 32-key half / which tile of the pass

    const bf16* Qb = p.q + (size_t)bh * N * 64;
    const bf16* Kb = p.k + (size_t)bh * N * 64;
    const bf16* Vb = p.v + (size_t)bh * N * 64;

    bf16x8 qf[PLANES][4];
    {
        const int qrow = p.q_off + min(q0 + qcol, NQ - 1);
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                qf[pl][s] = *reinterpret_cast<const bf16x8*>(Qb + (size_t)pl * p.qk_plane + (size_t)qrow * 64 + s * 16 + hh * 8);
    }
    // staging: 512 16-byte chunks per tile and plane (row = key, 8 chunks of 8 d), 2 per thread
    int st_row[2], st_chunk[2], st_koff[2], st_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 256;
        st_row[i] = idx >> 3;
        st_chunk[i] = idx & 7;
        st_koff[i] = lds_off128(st_row[i], st_chunk[i]);
        st_voff[i] = lds_off_v(st_row[i], st_chunk[i]);
    }
    const int nkt = (N + 63) / 64, npass = (nkt + 1) / 2;
    u32x4 rk[2][PLANES][2], rv[2][PLANES][2];
    auto load_pair = [&](int pass) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int kt = min(2 * pass + t, nkt - 1);  // (an odd tile count: the second slot of the last pass is never read)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const size_t off = (size_t)min(kt * 64 + st_row[i], N - 1) * 64 + st_chunk[i] * 8;  // rows past the end: finite values, P = 0 there
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) {
                    rk[t][pl][i] = *reinterpret_cast<const u32x4*>(Kb + (size_t)pl * p.qk_plane + off);
                    rv[t][pl][i] = *reinterpret_cast<const u32x4*>(Vb + (size_t)pl * p.qk_plane + off);
                }
            }
        }
    };
    auto store_pair = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) {
                    *reinterpret_cast<u32x4*>(smem + t * STAGE_BYTES + pl * TILE_BYTES + st_koff[i]) = rk[t][pl][i];
                    *reinterpret_cast<u32x4*>(smem + t * STAGE_BYTES + (PLANES + pl) * TILE_BYTES + st_voff[i]) = rv[t][pl][i];
                }
    };
    int k_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) k_off[s] = lds_off128(kb * 32 + qcol, 2 * s + hh);
    int v_base[2];
    {
        const int g = lane >> 4, q = (lane >> 2) & 3, pc = lane & 3;
#pragma unroll
        for (int db = 0; db < 2; ++db) v_base[db] = lds_off_v(4 * (g >> 1) + q, db * 4 + (g & 1) * 2 + (pc >> 1)) + (pc & 1) * 8;
    }

    f32x16 oacc[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    load_pair(0);
    store_pair();
    __syncthreads();
    for (int pass = 0; pass < npass; ++pass) {
        const int kt = 2 * pass + slot;
        if (kt < nkt) {
            const char* base = smem + slot * STAGE_BYTES;
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(base + k_off[s]);
                if constexpr (PLANES == 2) {
                    const bf16x8 kl = *reinterpret_cast<const bf16x8*>(base + TILE_BYTES + k_off[s]);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qf[0][s], sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[PLANES - 1][s], sacc, 0, 0, 0);
                }
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[0][s], sacc, 0, 0, 0);
            }
            bf16x4 vfr[2][2][PLANES][2];  // [k-step][d-block][plane][half]
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int pl = 0; pl < PLANES; ++pl) {
                        const char* vb = base + (PLANES + pl) * TILE_BYTES + v_base[db] + (kb * 2 + ks) * 2048;
                        vfr[ks][db][pl][0] = lds_read_tr16(vb);
                        vfr[ks][db][pl][1] = lds_read_tr16(vb + 1024);
                    }
            if (kt == nkt - 1 && (N & 63)) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh >= N) sacc[r] = -INFINITY;
            }
            float mx = sacc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
            mx = max_lane_xor32(mx);
            const float m_new = fmaxf(m_run, mx);  // (m_run >= -1e30 keeps an all-masked half finite: its P is exactly 0)
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * kLog2eT);
            m_run = m_new;
            const float mc = m_new * kLog2eT;
            float rowsum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[r], kLog2eT, -mc));
                sacc[r] = pv;
                rowsum += pv;
            }
            l_run = l_run * alpha + rowsum;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 ph, plo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pv = sacc[8 * ks + j];
                    const bf16 hi = (bf16)pv;
                    ph[j] = hi;
                    if constexpr (PLANES == 2) plo[j] = (bf16)(pv - (float)hi);
                }
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const bf16x8 vf = __builtin_shufflevector(vfr[ks][db][0][0], vfr[ks][db][0][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    if constexpr (PLANES == 2) {
                        const bf16x8 vl = __builtin_shufflevector(vfr[ks][db][PLANES - 1][0], vfr[ks][db][PLANES - 1][1], 0, 1, 2, 3, 4, 5, 6, 7);
                        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, oacc[db], 0, 0, 0);
                        oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, plo, oacc[db], 0, 0, 0);
                    }
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, ph, oacc[db], 0, 0, 0);
                }
            }
        }
        // (the next pair is requested only now: held in registers across the MFMA section it would cost the regular path of the
        // same kernel 10 VGPRs of allocation; the co-resident workgroup covers the latency)
        if (pass + 1 < npass) load_pair(pass + 1);
        __syncthreads();  // every wave is done reading the two slots
        if (pass + 1 < npass) {
            store_pair();
            __syncthreads();
        }
    }

    // ---- pairwise merge of the four partials through LDS: (2, 3) -> (0, 1), then 1 -> 0.  Lane i of every wave holds the same
    // (query column, d rows), so the merge is lane-wise: m = max, O and l scaled by exp2((m_x - m) log2 e) ----
    float* mbuf = reinterpret_cast<float*>(smem);  // [2 waves][34][64 lanes]
    auto put = [&](int w2) {
        float* d = mbuf + (size_t)w2 * 34 * 64 + lane;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) d[(db * 16 + r) * 64] = oacc[db][r];
        d[32 * 64] = m_run;
        d[33 * 64] = l_run;
    };
    auto take = [&](int w2) {
        const float* s = mbuf + (size_t)w2 * 34 * 64 + lane;
        const float m_o = s[32 * 64], l_o = s[33 * 64];
        const float m_new = fmaxf(m_run, m_o);
        const float fa = __builtin_amdgcn_exp2f((m_run - m_new) * kLog2eT), fb = __builtin_amdgcn_exp2f((m_o - m_new) * kLog2eT);
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[db][r] = fmaf(oacc[db][r], fa, s[(db * 16 + r) * 64] * fb);
        l_run = fmaf(l_run, fa, l_o * fb);
        m_run = m_new;
    };
    if (wave >= 2) put(wave - 2);
    __syncthreads();
    if (wave < 2) take(wave);
    __syncthreads();
    if (wave == 1) put(0);
    __syncthreads();
    if (wave != 0) return;
    take(0);

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int q = q0 + qcol;
    if (q < NQ) {
        const int64_t orow = (int64_t)b * NQ + q;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 hi4, lo4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = oacc[db][4 * g + e] * inv;
                    const bf16 hi = (bf16)v;
                    hi4[e] = hi;
                    if constexpr (PLANES == 2) lo4[e] = (bf16)(v - (float)hi);
                }
                const int d0 = db * 32 + 8 * g + 4 * hh;
                bf16* dst = p.o + a_pos<PLANES>(orow, p.ldo, h * 64 + d0);
                *reinterpret_cast<bf16x4*>(dst) = hi4;
                if constexpr (PLANES == 2) *reinterpret_cast<bf16x4*>(dst + kLoOffset) = lo4;
            }
    }
}

// true when workgroup (qt of nqb) is the ragged last tile and holds a single 32-row query block
__device__ __forceinline__ bool attention_is_split_tail(const AttnParams& p, int qt, int nqb, int NQ) {
    const int rows = NQ - qt * 128;
    return p.tail_split && nqb > 1 && qt == nqb - 1 && rows <= 32 && p.n_tok > 128;
}

}  // namespace cwm
