// Software-pipelined attention kernel (head_dim 64): the per-wave arithmetic of attention_kernel (attention.hip), with the
// matrix and vector work of consecutive key tiles interleaved inside each wave.
//
// PMC counters on attention_kernel (profiles/r1m_pmc_summary.json) show the matrix pipe busy ~50 % of the time and the
// waves issue-stalled for as long: within a wave the tile is a strict chain  S = K Q^T (MFMA) -> softmax (VALU) -> P V
// (MFMA), so MFMA and VALU work only overlap by chance between the two waves of a SIMD.  Per tile and wave the two
// pipes carry about the same load (parity mode: 48 MFMAs of 32 cycles vs ~1500 VALU cycles), so a wave has to keep
// both busy by itself.  Here iteration t of a wave runs, in instruction order,
//     phase A:  S(t+1) = K(t+1) Q^T         one MFMA per slot, each followed by a slice of softmax(t)
//     phase B:  O += V(t)^T P(t)^T          one MFMA per slot, each followed by a slice of the bf16 (hi, lo) split of
//                                           the next k-step's P
// with sched_barrier(0) between slots, so that the ~7 VALU issue cycles behind every MFMA issue are filled with
// independent work.  The two S accumulator chains (key halves) and the two O chains (d halves) alternate slot by slot: a
// dependent MFMA never issues straight behind its producer, and every accumulator still sees the additions in
// attention_kernel's order -- the outputs are bit-identical (tests/test_kernels_gpu.py).
//
// K / V tiles go global -> LDS by LDS-DMA, two slots each: iteration t first issues K(t+2) (slot of K(t), whose S was
// computed in iteration t - 1) and V(t+1) (slot of V(t-1)), and ends with vmcnt(0) + one workgroup barrier, a whole
// tile of work later.
//
// Round 5: the tile function takes its role (steady / second-to-last / last key tile) as a template argument and every runtime branch of a
// tile sits where no inline-asm LDS read is in flight (see BRANCHES below).  Measured and NOT kept (profiles/r5_attention_chain*.log, DESIGN.md
// section 4.10): work items that chain 2 - 6 query tiles of one head through the query-tile boundary (next tile's S(0) under the last key tile's
// softmax, LDS ring wrapping around, outputs stored under the next tile) -- bit-identical, the isolated batch-32 encoder launch 213 -> 201.5 us,
// the half-batch launches of a two-lane call +-0, the step +0.2 ... +1.2 % (two lanes) / -0.3 % (one lane).
#include "attention_tail.h"
#include <algorithm>
#include <cstdio>
#include <map>
#include <mutex>
#include <utility>

namespace cwm {

#ifdef CWM_ATTN_PROF
__device__ unsigned long long g_pipe_prof[8];
__device__ unsigned long long g_pipe_blocks[8192 * 8];
#define PROF_T(i) do { if (prof) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[i] += t_ - tlast; tlast = t_; } } while (0)
#else
#define PROF_T(i) do {} while (0)
#endif

namespace {

constexpr float kLog2e = 1.4426950408889634f;
constexpr int kKsplitMaxWgs = 512;  // workgroups of a key-split tail round (at most one round of slots on MI355X)

template <int PLANES>
struct PipeWave {
    bf16x8 qf[PLANES][4];
    f32x16 oacc[2];
    float m_run, l_run;
    int k_off[2][4];
    int hh;
};

// S = K Q^T for one tile without interleaving (prologue)
template <int PLANES>
__device__ __forceinline__ void qk_plain(const PipeWave<PLANES>& w, const char* kbase, f32x16 (&s)[2]) {
    constexpr int TILE_BYTES = 64 * 64 * 2;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kbase + w.k_off[kb][sx]);
            if constexpr (PLANES == 2) {
                const bf16x8 kl = *reinterpret_cast<const bf16x8*>(kbase + TILE_BYTES + w.k_off[kb][sx]);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, w.qf[0][sx], s[kb], 0, 0, 0);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, w.qf[1][sx], s[kb], 0, 0, 0);
            }
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, w.qf[0][sx], s[kb], 0, 0, 0);
        }
    }
}

// bf16 (hi, lo) split of the P values pair (2 j, 2 j + 1) of k-step KS
template <int PLANES>
__device__ __forceinline__ void split_pair(const f32x16 (&sc)[2], int ks, int j, bf16x8& ph, bf16x8& plo) {
#pragma unroll
    for (int e = 2 * j; e < 2 * j + 2; ++e) {
        const float pv = sc[ks >> 1][8 * (ks & 1) + e];
        const bf16 hi = (bf16)pv;
        ph[e] = hi;
        if constexpr (PLANES == 2) plo[e] = (bf16)(pv - (float)hi);
    }
}

// One k-step of O^T += V^T P^T: 2 NM slots (MFMA + a slice of the next k-step's P split)
template <int PLANES, int KS>
__device__ __forceinline__ void pv_step(PipeWave<PLANES>& w, const f32x16 (&sc)[2], const u32x2 (&vr)[2][PLANES][2], const bf16x8& ph, const bf16x8& plo,
                                        bf16x8& ph_next, bf16x8& plo_next) {
    constexpr int NM = PLANES == 2 ? 3 : 1;
    constexpr int NSLOT = 2 * NM;
    bf16x8 vf[2], vl[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        vf[db] = __builtin_shufflevector(__builtin_bit_cast(bf16x4, vr[db][0][0]), __builtin_bit_cast(bf16x4, vr[db][0][1]), 0, 1, 2, 3, 4, 5, 6, 7);
        if constexpr (PLANES == 2)
            vl[db] = __builtin_shufflevector(__builtin_bit_cast(bf16x4, vr[db][PLANES - 1][0]), __builtin_bit_cast(bf16x4, vr[db][PLANES - 1][1]), 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int slot = m * 2 + db;
            if constexpr (PLANES == 2) {
                if (m == 0) w.oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl[db], ph, w.oacc[db], 0, 0, 0);
                if (m == 1) w.oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[db], plo, w.oacc[db], 0, 0, 0);
                if (m == 2) w.oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[db], ph, w.oacc[db], 0, 0, 0);
            } else {
                w.oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[db], ph, w.oacc[db], 0, 0, 0);
            }
            if constexpr (KS < 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j * NSLOT / 4 == slot) split_pair<PLANES>(sc, KS + 1, j, ph_next, plo_next);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
}

// One key tile of one wave: phase A (S of the next tile into sn, softmax of this tile in sc), phase B (P V of this tile).
//   kbase: LDS image of the K tile of the NEXT S; va0 / va1: LDS addresses of this lane's V(t) transposed-read blocks (d halves)
//   dma(i): issues LDS-DMA piece i (< NDMA) of the tiles staged in this iteration -- one per slot at the head of phase
//   A: issued back to back they hold the wave for ~60 cycles each (measured: ~500 cycles per tile)
//   KIND (compile time; the control flow inside a tile must not depend on anything else -- see the note on branches below):
//     TILE_STEADY     next S = the next key tile of the same query tile
//     TILE_PENULT     the same, and the next key tile is the LAST of this workgroup's key range: if that is the sequence's last tile and it holds
//                     at most 32 keys (`half_next`; every model shape: N = 792, 1568, 3168 are 32 mod 64 or less) the S MFMAs of its second,
//                     all-padding key block are skipped (TILE_LAST sets those values to -inf anyway)
//     TILE_LAST       last key tile: sequence-end mask, no next S; k-steps 2 / 3 of P V (P exactly 0) skipped when `half_this`
//   Skipping is bit-identical: an accumulator that only ever adds products to +0 is never -0, so adding the +-0 products changes no bit.
//   BRANCHES: hipcc takes the output registers of the inline-asm V^T reads for written when the statement issues, so a control-flow merge
//   between such a read and its hand-counted wait may copy stale registers (round 4 shipped that race for a few hours).  The reads are issued
//   in the last two slots of phase A and inside phase B; every runtime branch of a tile therefore sits BEFORE slot NS - 2 (the DMA pieces, the
//   rescale, the half_next skip -- which leaves slot NS - 1's MFMA unconditional for that reason) or after a wait that leaves no read in
//   flight (`half_this`).  tools/asm_lds_lint.py fails on any branch or label with such reads pending (tests/test_host_logic.py).
enum : int { TILE_STEADY = 0, TILE_PENULT = 1, TILE_LAST = 2 };
template <int PLANES, int KIND, int NDMA, typename Dma>
__device__ __forceinline__ void pipe_tile(PipeWave<PLANES>& w, f32x16 (&sc)[2], f32x16 (&sn)[2], const char* kbase, unsigned va0, unsigned va1, int kt, int N, bool half_next,
                                          bool half_this, Dma&& dma
#ifdef CWM_ATTN_PROF
    , bool prof, unsigned long long (&pacc)[8], unsigned long long& tlast
#endif
    ) {
    constexpr bool HAS_NEXT = KIND != TILE_LAST;
    constexpr bool LASTK = KIND == TILE_LAST;
    constexpr int TILE_BYTES = 64 * 64 * 2;
    constexpr int NM = PLANES == 2 ? 3 : 1;
    constexpr int NS = 8 * NM;       // slots of phase A
    constexpr int NMAX = NS / 4;     // ... of which the first carry the running-maximum chain,
    constexpr int NE = NS - NMAX - 1;  // one the maximum's cross-lane step, the rest the exponentials
    constexpr int NRD = 4 * PLANES;  // transposed reads per k-step

    if constexpr (LASTK) {
        if (N & 63) {  // keys past the sequence end (last tile only)
            const int hh4 = 4 * w.hh;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 64 + kb * 32 + (r & 3) + 8 * (r >> 2) + hh4;
                    if (key >= N) sc[kb][r] = -INFINITY;
                }
        }
    }

    // ---- phase A --------------------------------------------------------------------------------------
    // slot order g = 2 sx + kb: the two accumulator chains alternate; K fragments are requested two groups ahead
    bf16x8 kfr[8][PLANES];
    auto read_k = [&](int g) {
        const int kb = g & 1, sx = g >> 1;
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl) kfr[g][pl] = *reinterpret_cast<const bf16x8*>(kbase + pl * TILE_BYTES + w.k_off[kb][sx]);
    };
    if constexpr (HAS_NEXT) {
        read_k(0);
        read_k(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sn[kb][r] = 0.f;
    }
    float mx = sc[0][0];
    float m_new = 0.f, alpha = 1.f, mc = 0.f, rowsum = 0.f;
    bool grew = false;
    u32x2 vr0[2][PLANES][2], vr1[2][PLANES][2];  // V^T fragments of phase B, k-steps 0 / 1 requested in the last two slots of phase A
    bf16x8 ph0, pl0, ph1, pl1;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int kb = g & 1, sx = g >> 1;
        if constexpr (HAS_NEXT) {
            if (g + 2 < 8) read_k(g + 2);
        }
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int slot = g * NM + m;
            if (slot < NDMA) dma(slot);
            // (half_next is a literal false everywhere but in TILE_PENULT; slot NS - 1 issues behind the first V^T reads: no branch there)
            if (HAS_NEXT && !(kb == 1 && half_next && slot < NS - 1)) {
                if constexpr (PLANES == 2) {
                    if (m == 0) sn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[g][PLANES - 1], w.qf[0][sx], sn[kb], 0, 0, 0);
                    if (m == 1) sn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[g][0], w.qf[PLANES - 1][sx], sn[kb], 0, 0, 0);
                    if (m == 2) sn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[g][0], w.qf[0][sx], sn[kb], 0, 0, 0);
                } else {
                    sn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[g][0], w.qf[0][sx], sn[kb], 0, 0, 0);
                }
            }
            if (slot < NMAX) {
#pragma unroll
                for (int v = 32 * slot / NMAX; v < 32 * (slot + 1) / NMAX; ++v) mx = fmaxf(mx, sc[v >> 4][v & 15]);
            } else if (slot == NMAX) {
                mx = max_lane_xor32(mx);
                m_new = fmaxf(w.m_run, mx);
                // the running maximum rarely moves after the first tiles: skip the rescale of O and l (alpha would be exactly 1)
                grew = __any(m_new > w.m_run);
                alpha = grew ? __builtin_amdgcn_exp2f((w.m_run - m_new) * kLog2e) : 1.0f;
                w.m_run = m_new;
                mc = m_new * kLog2e;
                // (rescale here, not between the phases: a branch there lets LLVM sink every exponential below it, out of
                // the MFMA shadow)
                // (the last tile has no MFMA slots to protect and the branch made hipcc keep a second copy of the 32 accumulator registers
                // alive across it -- 5 dwords of scratch: there the multiplication is unconditional, alpha = 1.0f exactly when nothing grew)
                if (KIND == TILE_LAST || grew) {
#pragma unroll
                    for (int db = 0; db < 2; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r) w.oacc[db][r] *= alpha;
                }
            } else {
                // exponentials of this slot's values; the row sum takes the previous slot's (an addition straight behind its
                // v_exp_f32 waits out the transcendental latency)
                const int j = slot - NMAX - 1;
                if (j > 0) {
#pragma unroll
                    for (int v = 32 * (j - 1) / NE; v < 32 * j / NE; ++v) rowsum += sc[v >> 4][v & 15];
                }
#pragma unroll
                for (int v = 32 * j / NE; v < 32 * (j + 1) / NE; ++v) sc[v >> 4][v & 15] = __builtin_amdgcn_exp2f(fmaf(sc[v >> 4][v & 15], kLog2e, -mc));
                asm volatile("" : "+v"(rowsum));  // keeps the additions in their slot (they otherwise sink behind the last MFMA)
            }
            // head start for phase B: the P values of k-step 0 were exponentiated long ago -- split one pair in each of the last four
            // slots -- and the V^T fragments of k-steps 0 / 1 are requested behind the last K-fragment wait
            if (slot >= NS - 4) split_pair<PLANES>(sc, 0, slot - (NS - 4), ph0, pl0);
            if (slot == NS - 2) lds_read_v_step<0, PLANES>(vr0, va0, va1);
            if (slot == NS - 1) lds_read_v_step<1, PLANES>(vr1, va0, va1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    PROF_T(1);
    // ---- phase B --------------------------------------------------------------------------------------
#pragma unroll
    for (int v = 32 * (NE - 1) / NE; v < 32; ++v) rowsum += sc[v >> 4][v & 15];
    w.l_run = w.l_run * alpha + rowsum;
    __builtin_amdgcn_sched_barrier(0);
    lds_wait_v_step<NRD, PLANES>(vr0);
    pv_step<PLANES, 0>(w, sc, vr0, ph0, pl0, ph1, pl1);
    if constexpr (LASTK) {
        // Last tile (once per query tile).  The branch on `half_this` must NOT sit between an asm LDS read and its wait: hipcc takes the asm's output registers
        // for written when the statement issues, and at a control-flow merge it is free to move them -- a first form of this code (branch right here, with
        // k-step 1's fragments still in flight) made it copy 16 fragment registers ~85 instructions after the reads and BEFORE the wait.  Right as long as
        // the LDS answers faster than that; beside another kernel's LDS traffic (two batch lanes) one wave in ~10^5 multiplied stale registers (ViT-L/4
        // batch 8: 5 % of the forwards wrong in one sample).  tools/asm_lds_lint.py checks the ISA for this (tests/test_host_logic.py).
        lds_wait_v_step<0, PLANES>(vr1);
        pv_step<PLANES, 1>(w, sc, vr1, ph1, pl1, ph0, pl0);
        if (!half_this) {
            lds_read_v_step<2, PLANES>(vr0, va0, va1);
            lds_read_v_step<3, PLANES>(vr1, va0, va1);
            lds_wait_v_step<NRD, PLANES>(vr0);
            pv_step<PLANES, 2>(w, sc, vr0, ph0, pl0, ph1, pl1);
            lds_wait_v_step<0, PLANES>(vr1);
            pv_step<PLANES, 3>(w, sc, vr1, ph1, pl1, ph0, pl0);
        }
    } else {
        lds_read_v_step<2, PLANES>(vr0, va0, va1);
        lds_wait_v_step<NRD, PLANES>(vr1);
        pv_step<PLANES, 1>(w, sc, vr1, ph1, pl1, ph0, pl0);
        lds_read_v_step<3, PLANES>(vr1, va0, va1);
        lds_wait_v_step<NRD, PLANES>(vr0);
        pv_step<PLANES, 2>(w, sc, vr0, ph0, pl0, ph1, pl1);
        lds_wait_v_step<0, PLANES>(vr1);
        pv_step<PLANES, 3>(w, sc, vr1, ph1, pl1, ph0, pl0);
    }
    PROF_T(2);
}

}  // namespace

template <int PLANES, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void attention_pipe_kernel(const AttnParams p) {
    constexpr int TILE_BYTES = 64 * 64 * 2;         // one 64x64 bf16 tile
    constexpr int SLOT_BYTES = TILE_BYTES * PLANES;  // hi [, lo] planes of one K or V tile
    constexpr int V_BASE = 2 * SLOT_BYTES;           // LDS: 2 K slots, then 2 V slots
    constexpr int NP = 8 * PLANES / NW;              // 1-KiB LDS-DMA pieces per wave, tile and operand
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void gbl_void;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qcol = lane & 31, hh = lane >> 5;
    const int N = p.n_tok;
    const int NQ = p.n_q > 0 ? p.n_q : N;
    const int nkt_all = (N + 63) / 64;
    // work item of this workgroup (1-D grid in dispatch order).  Key-split tail round: the items of a last round that would fill at most half of
    // the chip's workgroup slots are cut into ks_parts key ranges each, one workgroup per range, so that the round ends after 1 / ks_parts of a
    // workgroup period instead of a whole one (ViT-L/4 decoder, batch 8: 3136 items on 512 slots = 6.125 rounds -> 6 + 1/8)
    int item = blockIdx.x, kt0 = 0, nkt = nkt_all, part = -1;
    if (p.ks_parts > 0 && item >= p.ks_main) {
        const int sidx = item - p.ks_main;
        item = p.ks_main + sidx / p.ks_parts;
        part = sidx - (sidx / p.ks_parts) * p.ks_parts;
        kt0 = nkt_all * part / p.ks_parts;
        nkt = nkt_all * (part + 1) / p.ks_parts;  // this workgroup's key tiles: [kt0, nkt)
    }
    int qt, bh;
    attn_tile_of_item(item, p.ks_nqb, p.batch * p.heads, NQ, 32 * NW, p.remap != 0, qt, bh);
    if constexpr (NW == 4) {
        if (attention_is_split_tail(p, qt, p.ks_nqb, NQ)) {  // ragged last tile of <= 32 rows: the four waves split the keys (attention_tail.h)
            attention_tail_block<PLANES>(p, smem, bh, qt * 128, NQ);
            return;
        }
    }
    const int b = bh / p.heads, h = bh - b * p.heads;
    const int q0 = qt * (32 * NW) + wave * 32;
    const bool active = q0 < NQ;  // idle waves (query rows past the end) only stage tiles and keep the barrier count

    const bf16* Qb = p.q + (size_t)bh * N * 64;
    const bf16* Kb = p.k + (size_t)bh * N * 64;
    const bf16* Vb = p.v + (size_t)bh * N * 64;

    PipeWave<PLANES> w;
    w.hh = hh;
    {
        const int qrow = p.q_off + min(q0 + qcol, NQ - 1);
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                w.qf[pl][s] = *reinterpret_cast<const bf16x8*>(Qb + (size_t)pl * p.qk_plane + (size_t)qrow * 64 + s * 16 + hh * 8);
    }

    // ---- LDS-DMA bookkeeping: piece = 8 key rows x 128 B; lane l lands at chunk l % 8 of row l / 8 ----
    int st_lds[NP], st_koff[NP], st_voff[NP];  // LDS offset of the piece; byte offset of the lane's 16-byte chunk inside a K / V tile
    const char *st_kp[NP], *st_vp[NP];         // wave-uniform plane bases: the DMA address is base (SGPR pair) + 32-bit lane offset
#pragma unroll
    for (int jj = 0; jj < NP; ++jj) {
        const int pi = wave * NP + jj, plane = pi >> 3, pc = pi & 7;
        const int r = pc * 8 + (lane >> 3);
        st_lds[jj] = plane * TILE_BYTES + pc * 1024;
        st_koff[jj] = r * 128 + ((lane & 7) ^ ((r >> 1) & 7)) * 16;         // K image: chunk c of row r at c ^ ((r >> 1) & 7)
        st_voff[jj] = r * 128 + ((lane & 7) ^ (((r >> 1) & 1) << 2)) * 16;  // V image: lds_off_v
        st_kp[jj] = reinterpret_cast<const char*>(Kb + (size_t)plane * p.qk_plane);
        st_vp[jj] = reinterpret_cast<const char*>(Vb + (size_t)plane * p.qk_plane);
    }
    // CLAMP: the tile may be the last one of a ragged sequence -- rows past the end re-read the last key (P is exactly 0 there).
    // The steady-state loop stages tiles that cannot be the last one and pays a single v_add per piece.
    auto tile_off = [&](int kt, int toff, auto clamp_c) -> unsigned {
        if constexpr (decltype(clamp_c)::value) return (unsigned)(min(kt * 64 + (toff >> 7), N - 1) * 128 + (toff & 127));
        else return (unsigned)(kt * 8192 + toff);
    };
    // key tile kt -> ring slot `slot` (0 / 1)
    auto stage_k1 = [&](int kt, int slot, int jj, auto clamp_c) {
        __builtin_amdgcn_global_load_lds((gbl_void*)(st_kp[jj] + tile_off(kt, st_koff[jj], clamp_c)), (lds_void*)(smem + slot * SLOT_BYTES + st_lds[jj]), 16, 0, 0);
    };
    auto stage_v1 = [&](int kt, int slot, int jj, auto clamp_c) {
        __builtin_amdgcn_global_load_lds((gbl_void*)(st_vp[jj] + tile_off(kt, st_voff[jj], clamp_c)), (lds_void*)(smem + V_BASE + slot * SLOT_BYTES + st_lds[jj]), 16, 0,
                                         0);
    };
    auto stage_k = [&](int kt, int slot) {
#pragma unroll
        for (int jj = 0; jj < NP; ++jj) stage_k1(kt, slot, jj, std::true_type{});
    };
    auto stage_v = [&](int kt, int slot) {
#pragma unroll
        for (int jj = 0; jj < NP; ++jj) stage_v1(kt, slot, jj, std::true_type{});
    };

#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s = 0; s < 4; ++s) w.k_off[kb][s] = lds_off128(kb * 32 + qcol, 2 * s + hh);
    unsigned v_addr[2];  // LDS addresses of this lane's transposed-read blocks in V slot 0
    {
        const int g = lane >> 4, q = (lane >> 2) & 3, pc = lane & 3;
#pragma unroll
        for (int db = 0; db < 2; ++db)
            v_addr[db] = (unsigned)(size_t)(lds_void*)(smem + V_BASE + lds_off_v(4 * (g >> 1) + q, db * 4 + (g & 1) * 2 + (pc >> 1)) + (pc & 1) * 8);
    }

#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) w.oacc[db][r] = 0.f;
    w.m_run = -1e30f;
    w.l_run = 0.f;

#define CWM_TILE_END()                                     \
    do {                                                   \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   \
        __builtin_amdgcn_sched_barrier(0);                 \
        __builtin_amdgcn_s_barrier();                      \
        __builtin_amdgcn_sched_barrier(0);                 \
    } while (0)
// key tile KT: S accumulators SC (this tile) -> SN (next tile).  Ring slots: tile kt sits in slot kt & 1; this iteration re-fills the K slot of
// THIS tile (its S is done) with the tile two ahead and the other V slot with the tile one ahead.
#define CWM_TILE(KIND, SC, SN, KT, CLAMP)                                                                                              \
    do {                                                                                                                               \
        const int kt_ = (KT);                                                                                                          \
        const int ks_ = kt_ & 1;                                                                                                       \
        auto dma = [&](int i) {                                                                                                        \
            if (i < NP) {                                                                                                              \
                if ((KIND) == TILE_STEADY) stage_k1(kt_ + 2, ks_, i, CLAMP);                                                           \
            } else if ((KIND) != TILE_LAST) {                                                                                          \
                stage_v1(kt_ + 1, ks_ ^ 1, i - NP, CLAMP);                                                                             \
            }                                                                                                                          \
        };                                                                                                                             \
        PROF_T(0);                                                                                                                     \
        if (active) {                                                                                                                  \
            pipe_tile<PLANES, KIND, 2 * NP>(w, SC, SN, smem + (ks_ ^ 1) * SLOT_BYTES, v_addr[0] + ks_ * SLOT_BYTES, v_addr[1] + ks_ * SLOT_BYTES, kt_, N,  \
                                            (KIND) == TILE_PENULT && half_last, half_last, dma PROF_ARGS);                             \
        } else {                                                                                                                       \
            if ((KIND) == TILE_STEADY) stage_k(kt_ + 2, ks_);                                                                          \
            if ((KIND) != TILE_LAST) stage_v(kt_ + 1, ks_ ^ 1);                                                                        \
        }                                                                                                                              \
        if ((KIND) != TILE_LAST) CWM_TILE_END();                                                                                       \
        PROF_T(3);                                                                                                                     \
    } while (0)

#ifdef CWM_ATTN_PROF
#define PROF_ARGS , prof, pacc, tlast
    const bool prof = wave == 0;  // every workgroup's wave 0 keeps phase totals (written to g_pipe_blocks)
    unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = tlast, r_begin = __builtin_amdgcn_s_memrealtime();
#else
#define PROF_ARGS
#endif
    // the sequence's last key tile is in this workgroup's range and at most half full (pipe_tile: half_next / half_this)
    const bool half_last = nkt == nkt_all && ((N - 1) & 63) < 32;

    // ---- normalise and store O[q][h*64 + d] (or, key-split tail round: leave (O^T unnormalised, running max, sum) of this key range for
    // attention_combine_kernel) ----
    auto store_out = [&]() {
        const float l_tot = w.l_run + __shfl_xor(w.l_run, 32, 64);
        const int q = q0 + qcol;
        if (part >= 0) {
            if (q < NQ) {
                float* rec = p.ks_scratch + (((size_t)(item - p.ks_main) * p.ks_parts + part) * 128 + (wave * 32 + qcol)) * 68;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<f32x4*>(rec + db * 32 + 8 * g + 4 * hh) =
                            f32x4{w.oacc[db][4 * g], w.oacc[db][4 * g + 1], w.oacc[db][4 * g + 2], w.oacc[db][4 * g + 3]};
                if (hh == 0) {
                    rec[64] = w.m_run;
                    rec[65] = l_tot;
                }
            }
            return;
        }
        const float inv = 1.0f / l_tot;
        if (q < NQ) {
            const int64_t orow = (int64_t)b * NQ + q;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 hi4, lo4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = w.oacc[db][4 * g + e] * inv;
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
    };

    // ---- prologue: K(kt0), V(kt0), K(kt0 + 1) staged, S(kt0) without interleaving ----
    f32x16 sa[2], sb[2];
    stage_k(kt0, kt0 & 1);
    stage_v(kt0, kt0 & 1);
    if (kt0 + 1 < nkt) stage_k(kt0 + 1, (kt0 + 1) & 1);
    CWM_TILE_END();
    if (active) qk_plain<PLANES>(w, smem + (kt0 & 1) * SLOT_BYTES, sa);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();  // K(kt0) is re-staged by the first tile
    __builtin_amdgcn_sched_barrier(0);

    // ---- key tiles; S of the current one is in sa ----
    // Tile instances: two steady ones (no row clamp: the tiles they stage are never the last one), one clamped steady tile for the 1 - 2 tiles
    // between the pair loop and the end, the second-to-last tile and the last one.  ONE instance per role, each with fixed S registers
    // (steady pairs sa -> sb -> sa, clamped sa -> sb, second-to-last sb -> sa, last sa): with two instances of a role -- S in sa after an odd
    // count, in sb after an even one -- hipcc merged the paths through copies of the S and O accumulators and spilled around them (round 4).
    // The parity is fixed up by moving S (32 v_mov) -- never for an odd tile count, which every model shape has (13, 25, 49 ... key tiles).
    int kt = kt0;
    const int n_mine = nkt - kt0;
    if (n_mine >= 3) {
        for (; kt + 4 < nkt; kt += 2) {
            CWM_TILE(TILE_STEADY, sa, sb, kt, std::false_type{});
            CWM_TILE(TILE_STEADY, sb, sa, kt + 1, std::false_type{});
        }
        for (;;) {  // 3 or 4 tiles left: one or two clamped steady tiles
            CWM_TILE(TILE_STEADY, sa, sb, kt, std::true_type{});
            ++kt;
            if (kt + 2 == nkt) break;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) sa[kb] = sb[kb];
        }
    } else if (n_mine == 2) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) sb[kb] = sa[kb];
    }
    if (n_mine >= 2) {
        CWM_TILE(TILE_PENULT, sb, sa, kt, std::true_type{});
        ++kt;
    }
    CWM_TILE(TILE_LAST, sa, sb, kt, std::true_type{});
    store_out();
#undef CWM_TILE
#undef CWM_TILE_END
#ifdef CWM_ATTN_PROF
    pacc[4] = __builtin_amdgcn_s_memtime() - t_begin;
    pacc[5] = __builtin_amdgcn_s_memrealtime() - r_begin;
    {
        const int bid = blockIdx.x;
        if (wave == 0 && lane == 0 && bid < 8192) {
            g_pipe_blocks[bid * 8 + 0] = r_begin;
            g_pipe_blocks[bid * 8 + 1] = r_begin + pacc[5];
            g_pipe_blocks[bid * 8 + 2] = pacc[4];
            g_pipe_blocks[bid * 8 + 3] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
            for (int i = 0; i < 4; ++i) g_pipe_blocks[bid * 8 + 4 + i] = pacc[i];
        }
    }
    if (prof && lane == 0 && blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = 0; i < 8; ++i) g_pipe_prof[i] = pacc[i];
#endif
}

#ifdef CWM_ATTN_PROF
int attention_pipe_prof(int i) {
    if (i >= 1000) {  // dump the per-block records to /tmp/attn_blocks.bin
        static unsigned long long h[8192 * 8];
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pipe_blocks), sizeof(h)) != hipSuccess) return -1;
        FILE* f = fopen("/tmp/attn_blocks.bin", "wb");
        if (!f) return -1;
        fwrite(h, 1, sizeof(h), f);
        fclose(f);
        return 0;
    }
    unsigned long long h[8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pipe_prof), sizeof(h)) != hipSuccess) return -1;
    return (int)h[i & 7];
}
#else
int attention_pipe_prof(int) { return -1; }
#endif

// Merge of the key ranges of the split items: one wave per query row, lane d.  O = sum_j O_j e^(m_j - m) / sum_j l_j e^(m_j - m), parts in
// ascending order (deterministic); then the bf16 (hi, lo) split and the store of attention_pipe_kernel's epilogue.
template <int PLANES>
__global__ __launch_bounds__(256) void attention_combine_kernel(const AttnParams p) {
    const int wave = threadIdx.x >> 6, d = threadIdx.x & 63;
    const int N = p.n_tok, NQ = p.n_q > 0 ? p.n_q : N;
    const int sitem = blockIdx.x / 32, qrow = (blockIdx.x % 32) * 4 + wave;
    int qt, bh;
    attn_tile_of_item(p.ks_main + sitem, p.ks_nqb, p.batch * p.heads, NQ, 128, p.remap != 0, qt, bh);
    const int q = qt * 128 + qrow;
    if (q >= NQ) return;
    const float* rec = p.ks_scratch + (((size_t)sitem * p.ks_parts) * 128 + qrow) * 68;
    float m = -INFINITY;
    for (int j = 0; j < p.ks_parts; ++j) m = fmaxf(m, rec[(size_t)j * 128 * 68 + 64]);
    float o = 0.f, l = 0.f;
    for (int j = 0; j < p.ks_parts; ++j) {
        const float* r = rec + (size_t)j * 128 * 68;
        const float wgt = __builtin_amdgcn_exp2f((r[64] - m) * kLog2e);
        o = fmaf(r[d], wgt, o);
        l = fmaf(r[65], wgt, l);
    }
    const float v = o / l;
    const int b = bh / p.heads, h = bh - b * p.heads;
    bf16* dst = p.o + a_pos<PLANES>((int64_t)b * NQ + q, p.ldo, h * 64 + d);
    const bf16 hi = (bf16)v;
    *dst = hi;
    if constexpr (PLANES == 2) dst[kLoOffset] = (bf16)(v - (float)hi);
}

// scratch of the key-split tail round: at most one round of partial workgroups (512 x 128 queries x 68 floats = 17.8 MB), one buffer per
// (device, stream) -- the batch lanes launch concurrently --, created on first need
static int ksplit_scratch(hipStream_t stream, float** out) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, float*> table;
    int dev = 0;
    CWM_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    auto it = table.find(std::make_pair(dev, stream));
    if (it == table.end()) {
        float* buf = nullptr;
        CWM_HIP_CHECK(hipMalloc((void**)&buf, (size_t)kKsplitMaxWgs * 128 * 68 * sizeof(float)));
        it = table.emplace(std::make_pair(dev, stream), buf).first;
    }
    *out = it->second;
    return 0;
}

template <int PLANES, int NW>
static int launch_pipe(const AttnParams& p_in, hipStream_t stream) {
    AttnParams p = p_in;
    const int nq = p.n_q > 0 ? p.n_q : p.n_tok;
    const int nqb = (nq + 32 * NW - 1) / (32 * NW), nbh = p.batch * p.heads;
    const int64_t n_items = (int64_t)nqb * nbh;
    CWM_REQUIRE(n_items < (1ll << 30), "attention: too many (query tile, batch, head) work items");
    p.ks_nqb = nqb;
    p.ks_main = (int)n_items;
    p.ks_parts = 0;
    p.ks_scratch = nullptr;
    int extra = 0;
    if (NW == 4 && (p.tune ? p.tune->attn_ksplit : 1)) {
        // a last round that fills at most a quarter of the slots (two workgroups per CU) costs most of a workgroup period for a fraction of a
        // round's work: split its items' keys.  Measured (profiles/r4_microbench_attn_ksplit.log): ViT-L/4 decoder (3136 items = 6 rounds + 64,
        // 8 ranges) 1646 -> 1597 us parity, 766 -> 753 us fast; a last round of 128 items in 4 ranges (ViT-L/4 encoder) gains nothing -- the
        // rounds are not in lockstep, so a tail costs less than the slot arithmetic says --, hence the quarter.  Not together with the
        // ragged-tile key split of attention_tail.h (those light tiles already fill the tail).
        const int slots = 2 * gemm_cu_count(), nkt = (p.n_tok + 63) / 64;
        const int rem = (int)(n_items % slots);
        const int last_rows = nq - (nqb - 1) * 128;
        const bool ragged_split = p.tail_split && nqb > 1 && last_rows <= 32 && p.n_tok > 128;
        if (n_items >= slots && rem > 0 && rem * 4 <= slots && !ragged_split) {
            int parts = std::min(8, slots / rem);
            while (parts > 1 && nkt / parts < 6) --parts;   // every key range keeps >= 6 tiles: its prologue and the merge stay small
            if (parts >= 2 && rem * parts <= kKsplitMaxWgs) {
                if (int rc = ksplit_scratch(stream, &p.ks_scratch)) return rc;
                p.ks_main = (int)(n_items - rem);
                p.ks_parts = parts;
                extra = rem * parts - rem;
            }
        }
    }
    const unsigned grid = (unsigned)(n_items + extra);
    const size_t smem = (size_t)4 * (64 * 64 * 2) * PLANES;  // 2 K slots + 2 V slots
    if (smem > 48 * 1024)
        if (int rc = cwm_set_max_lds((const void*)attention_pipe_kernel<PLANES, NW>, (int)smem)) return rc;
    hipLaunchKernelGGL((attention_pipe_kernel<PLANES, NW>), dim3(grid), dim3(64 * NW), smem, stream, p);
    if (p.ks_parts > 0)
        hipLaunchKernelGGL((attention_combine_kernel<PLANES>), dim3((unsigned)((n_items - p.ks_main) * 32)), dim3(256), 0, stream, p);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

// 4 waves: 128 queries per workgroup, two workgroups per CU.  (8 waves -- 256 queries, one workgroup per CU, half the
// LDS-DMA work per query -- measured 5-15 % slower: the barrier then couples all eight waves of the CU.)
int launch_attention_pipe(const AttnParams& p, int planes, hipStream_t stream) {
    return planes == 1 ? launch_pipe<1, 4>(p, stream) : launch_pipe<2, 4>(p, stream);
}

}  // namespace cwm
