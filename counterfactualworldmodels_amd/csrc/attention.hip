// Flash-style multi-head self-attention for CDNA4 (gfx950), head_dim = 64, non-causal, any N.
//
// Replaces the materialised  softmax(q k^T) v  of `Attention.forward`
// (reference: cwm/models/VideoMAE/utils.py:108-113; the optional CUDA flash-attn import at :71-73).
//
// Layout / dataflow (one workgroup = 4 waves = 128 query rows of one (batch, head); a wave owns 32 rows):
//   S^T[key][q] = K[key][:] . Q[q][:]       mfma_f32_32x32x16_bf16, A = K tile (LDS), B = Q (registers)
//   -> every lane holds 16 keys of ONE query column, so the online-softmax row max / row sum are
//      in-register reductions plus a single lane^32 exchange (wave shuffle)
//   O^T[d][q]  += V^T[d][key] . P^T[key][q]  A = V^T fragments, B = the S^T accumulator itself,
//      converted to bf16 in place (accumulator-as-operand; k order permuted, matched on the V reads)
//   -> the running rescale factor alpha[q] is lane-local as well.
// V is ROW-major in memory and in LDS ([key][d], like K -- the QKV projection writes Q, K and V with one
// epilogue); the V^T fragments come from the hardware transpose read ds_read_b64_tr_b16: per 16-lane group a
// block of 4 keys x 16 d is delivered column-major, i.e. a lane gets 4 consecutive keys of ONE d -- exactly one
// half of the permuted k order {4 hh + 0..3, 8 + 4 hh + 0..3} the accumulator-as-operand trick asks for.
// K and V tiles (64 keys) are double-buffered in LDS with register-staged prefetch of tile t+1
// issued before the MFMAs of tile t.  LDS images are XOR-swizzled (K: conflict-free ds_read_b128; V: the four
// key rows of a transposed read land on the four 64-byte quarters of the 256-byte bank row).
// PLANES==2 is the split-bf16 "parity" mode (hi*hi + hi*lo + lo*hi for both products).
#include "attention_tail.h"

namespace cwm {

template <int PLANES>
__global__ __launch_bounds__(256, 2) void attention_kernel(const AttnParams p) {
    constexpr int TILE_BYTES = 64 * 64 * 2;              // one 64x64 bf16 tile
    constexpr int STAGE_BYTES = TILE_BYTES * 2 * PLANES;  // K planes, then V^T planes
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qcol = lane & 31, hh = lane >> 5;
    const int N = p.n_tok;
    int qt, bh;
    attn_tile_of_block(p.n_q > 0 ? p.n_q : p.n_tok, 128, p.remap != 0, qt, bh);
    if (attention_is_split_tail(p, qt, gridDim.x, p.n_q > 0 ? p.n_q : p.n_tok)) {  // ragged last tile of <= 32 rows: the four waves split the keys
        attention_tail_block<PLANES>(p, smem, bh, qt * 128, p.n_q > 0 ? p.n_q : p.n_tok);
        return;
    }
    const int b = bh / p.heads, h = bh - b * p.heads;
    const int q0 = qt * 128 + wave * 32;
    // a wave whose 32 query rows all lie past the sequence end (N = 792: three of the 28 wave slots per head) only helps to
    // stage the K / V tiles: its MFMA / softmax work is skipped, leaving the matrix pipe to the co-resident workgroup
    const int NQ = p.n_q > 0 ? p.n_q : p.n_tok;  // queries are rows [q_off, q_off + NQ) (last decoder block: the masked tokens only)
    const bool active = q0 < NQ;

    const bf16* Qb = p.q + (size_t)bh * N * 64;
    const bf16* Kb = p.k + (size_t)bh * N * 64;
    const bf16* Vb = p.v + (size_t)bh * N * 64;

    // ---- Q fragments (B operand): lane (q = qcol, half hh) holds Q[q][16 s + 8 hh + 0..7] --------
    bf16x8 qf[PLANES][4];
    {
        const int qrow = p.q_off + min(q0 + qcol, NQ - 1);
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                qf[pl][s] = *reinterpret_cast<const bf16x8*>(Qb + (size_t)pl * p.qk_plane + (size_t)qrow * 64 + s * 16 + hh * 8);
    }

    // ---- staging bookkeeping: 512 16-byte chunks per tile (row = key, 8 chunks of 8 d), 2 per thread ----
    int st_row[2], st_chunk[2], st_koff[2], st_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 256;
        st_row[i] = idx >> 3;
        st_chunk[i] = idx & 7;
        st_koff[i] = lds_off128(st_row[i], st_chunk[i]);
        st_voff[i] = lds_off_v(st_row[i], st_chunk[i]);
    }
    u32x4 rk[PLANES][2], rv[PLANES][2];
    auto load_tiles = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // keys past the sequence end re-read the last row: finite values, and P is exactly 0 there
            const size_t off = (size_t)min(kt * 64 + st_row[i], N - 1) * 64 + st_chunk[i] * 8;
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) {
                rk[pl][i] = *reinterpret_cast<const u32x4*>(Kb + (size_t)pl * p.qk_plane + off);
                rv[pl][i] = *reinterpret_cast<const u32x4*>(Vb + (size_t)pl * p.qk_plane + off);
            }
        }
    };
    auto store_tiles = [&](int stage) {
        char* base = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl) {
                *reinterpret_cast<u32x4*>(base + pl * TILE_BYTES + st_koff[i]) = rk[pl][i];
                *reinterpret_cast<u32x4*>(base + (PLANES + pl) * TILE_BYTES + st_voff[i]) = rv[pl][i];
            }
    };

    // ---- fragment read offsets -------------------------------------------------------------------
    // K (A operand of S^T): row = key kb*32 + qcol, 16-byte chunk = 2 s + hh
    int k_off[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s = 0; s < 4; ++s) k_off[kb][s] = lds_off128(kb * 32 + qcol, 2 * s + hh);
    // V^T (A operand of O^T) by transposed reads.  16-lane group g = lane >> 4 reads the block {keys 4 (g >> 1) + 0..3
    // (+ 16 ks, + 8 for the second half of the fragment)} x {d = 32 db + 16 (g & 1) + 0..15}: lane 4 q + pc of the group
    // supplies the address of key row q, d columns 4 pc .. 4 pc + 3, and lane i receives d column i (= 32 db + lane % 32)
    // with key q in element q.  Key offsets 16 ks + 8 half are multiples of 4, so the swizzle bit is (q >> 1) & 1 and
    // they are plain immediates; the two db blocks differ by the swizzled chunk bit -> one base register each.
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
    const float kLog2e = 1.4426950408889634f;

    const int nkt = (N + 63) / 64;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    // One key tile.  LAST (peeled): no next tile to stage, and keys past the sequence end are masked -- only there, so that
    // the 32 selects per tile that the mask costs are not paid in every tile (the fast-mode loop is VALU-bound).
    auto tile_step = [&](int kt, auto last_c) {
        constexpr bool LAST = decltype(last_c)::value;
        const int cur = kt & 1;
        if constexpr (!LAST) load_tiles(kt + 1);
        const char* base = smem + cur * STAGE_BYTES;

        if (active) {
        // ---- S^T = K Q^T ------------------------------------------------------------------------
        // All K fragments of the tile are requested before the first MFMA (hipcc otherwise sinks every ds_read next to its
        // use: one exposed LDS latency per 3 MFMAs), and all V^T fragments right after the S MFMAs have been issued, so that
        // they land under the softmax's VALU work.
        bf16x8 kfr[2][4][PLANES];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) kfr[kb][s][pl] = *reinterpret_cast<const bf16x8*>(base + pl * TILE_BYTES + k_off[kb][s]);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if constexpr (PLANES == 2) {
                    sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kb][s][PLANES - 1], qf[0][s], sacc[kb], 0, 0, 0);
                    sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kb][s][0], qf[1][s], sacc[kb], 0, 0, 0);
                }
                sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[kb][s][0], qf[0][s], sacc[kb], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        bf16x4 vfr[4][2][PLANES][2];  // [k-step][d-block][plane][half]: transposed reads
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl) {
                    const char* vb = base + (PLANES + pl) * TILE_BYTES + v_base[db] + ks * 2048;
                    vfr[ks][db][pl][0] = lds_read_tr16(vb);
                    vfr[ks][db][pl][1] = lds_read_tr16(vb + 1024);
                }
        __builtin_amdgcn_sched_barrier(0);

        // ---- online softmax (per query column; lanes l and l^32 share a query) ------------------
        if constexpr (LAST) {
            if (N & 63) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kt * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        if (key >= N) sacc[kb][r] = -INFINITY;
                    }
            }
        }
        float mx = sacc[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
        mx = max_lane_xor32(mx);
        const float m_new = fmaxf(m_run, mx);
        // the running maximum rarely moves after the first tiles: skip the rescale of O and l (alpha would be exactly 1)
        const bool grew = __any(m_new > m_run);
        const float alpha = grew ? __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e) : 1.0f;
        m_run = m_new;
        const float mc = m_new * kLog2e;
        float rowsum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], kLog2e, -mc));
                sacc[kb][r] = pv;
                rowsum += pv;
            }
        l_run = l_run * alpha + rowsum;
        if (grew) {
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
        }

        // ---- O^T += V^T P^T  (P^T fragment of k-step ks = kb*2+s is sacc[kb][8s .. 8s+7]) --------
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kb = ks >> 1, s = ks & 1;
            bf16x8 ph, plo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float pv = sacc[kb][8 * s + j];
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
        }  // active

        if constexpr (!LAST) {
            store_tiles(cur ^ 1);
            __syncthreads();
        }
    };
    for (int kt = 0; kt + 1 < nkt; ++kt) tile_step(kt, std::false_type{});
    tile_step(nkt - 1, std::true_type{});

    // ---- normalise and store O[q][h*64 + d] ----------------------------------------------------------
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
                bf16* dst = p.o + a_pos<PLANES>(orow, p.ldo, h * 64 + d0);  // GEMM A-operand layout (common.h)
                *reinterpret_cast<bf16x4*>(dst) = hi4;
                if constexpr (PLANES == 2) *reinterpret_cast<bf16x4*>(dst + kLoOffset) = lo4;
            }
    }
}

// ---------------------------------------------------------------------------------------------------
int launch_attention(const AttnParams& p_in, int planes, hipStream_t stream) {
    AttnParams p = p_in;
    const Tuning& tn = p.tune ? *p.tune : default_tuning();
    p.remap = tn.attn_remap;
    p.tail_split = tn.attn_tail;
    CWM_REQUIRE(planes == 1 || planes == 2, "attention: planes must be 1 or 2");
    CWM_REQUIRE(p.n_tok > 0 && p.batch > 0 && p.heads > 0, "attention: empty problem");
    CWM_REQUIRE(p.ldo % 4 == 0, "attention: ldo must be a multiple of 4");
    CWM_REQUIRE((int64_t)p.n_tok * 128 < (1ll << 31), "attention: %d tokens exceed the 32-bit tile offsets within one (batch, head)", p.n_tok);
    CWM_REQUIRE(p.q_off >= 0 && p.n_q >= 0 && p.q_off + p.n_q <= p.n_tok, "attention: query rows [%d, %d) outside the %d tokens", p.q_off, p.q_off + p.n_q, p.n_tok);
    const int nqb = ((p.n_q > 0 ? p.n_q : p.n_tok) + 127) / 128;
    // Measured (tools/microbench.py attn_sweep / attn, MI355X, round 6: profiles/r6_microbench_attn_sweep.log, r6_microbench_attn.log -- 12 heads, ~2.4e8 score
    // elements per launch at every length):
    //   parity mode: the software-pipelined kernel (3) -- each wave overlaps the MFMAs of one tile with the softmax of the previous one -- is ahead from ~200 tokens
    //     (196: 273 vs 280 us; 792: 203 vs 229; 1568: 178 vs 201; 6272: 1599 vs 1834), the 4-wave kernel (1) below (128: 114 vs 122; 40: 41.6 vs 43.3);
    //   fast mode: one MFMA per product leaves the loop VALU-bound and the crossover sits higher: kernel 1 ahead up to ~400 tokens (256: 133 vs 146 us; 392: 141 vs 142),
    //     a tie at 512 (102.3 vs 102.2), kernel 3 ahead from there (792: 106.7 vs 109.7; 1568: 88 vs 99; 6272: 756 vs 840).
    //   Every sequence of the three model families (792 ... 6336 tokens) therefore runs kernel 3 in both modes; kernel 1 serves short sequences (test-sized grids).
    //   (Rounds 1-5 kept kernel 1 in fast mode below 2048 tokens on the strength of a round-1 log that later kernel-3 work had overtaken: 2 % of the fast-mode step.
    //   A staggered 8-wave kernel, "2", tied kernel 1 in parity mode and lost 25-35 % in fast mode: removed in round 4.)
    // Both produce bit-identical outputs (tests/test_kernels_gpu.py); Tuning.attn_kernel forces one.
    const int kern = tn.attn_kernel ? tn.attn_kernel : (p.n_tok >= (planes == 2 ? 160 : 512) ? 3 : 1);
    if (kern == 3) return launch_attention_pipe(p, planes, stream);
    CWM_REQUIRE(kern == 1, "attention: unknown kernel %d (1: 4-wave, 3: software-pipelined)", kern);
    const dim3 grid(nqb, p.batch * p.heads);
    const size_t smem = (size_t)2 * (64 * 64 * 2) * 2 * planes;
    if (planes == 1) {
        hipLaunchKernelGGL(attention_kernel<1>, grid, dim3(256), smem, stream, p);
    } else {
        if (int rc = cwm_set_max_lds((const void*)attention_kernel<2>, (int)smem)) return rc;
        hipLaunchKernelGGL(attention_kernel<2>, grid, dim3(256), smem, stream, p);
    }
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace cwm
