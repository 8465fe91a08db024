// bf16 / split-bf16 ("bf16x3") MFMA GEMM for CDNA4 (gfx950):  C[M,N] = A[M,K] * W[N,K]^T  (+ epilogue)
//
// Replaces every `F.linear` on the predictor path (reference call sites: VideoMAE/utils.py:48-53
// fc1/fc2, :93 qkv, :119 proj; vmae.py:547 encoder_to_decoder, :251 head; the Conv3d patch embed of
// VideoMAE/utils.py:174-197 expressed as an im2col GEMM).
//
// Kernels in this file (launch_gemm picks per shape, gemm_choose_tile):
//   gemm_bf16_kernel<PLANES, BM, BN, WM, WN>  one barrier + vmcnt(0) per K tile, 2-stage LDS ring; used as 128x128 / 256 threads
//                                            with two workgroups per CU for narrow outputs (N <= 768) and remainders
//   gemm8p_kernel<PLANES>                     256x256 / 512 threads, 8-phase main loop (staggered wave groups, LDS-DMA in flight
//                                            across raw barriers): wide outputs (N >= 1024) and whole rounds of narrow ones
// Common to both:
// * A and W are both K-contiguous, so one lane's MFMA fragment is one 16-byte LDS read
// * K tiles are staged by LDS-DMA (`global_load_lds_dwordx4`, 1 KiB per wave-instruction, no staging
//   VGPRs, no ds_write pass).  An LDS-DMA piece lands linearly (lane l -> base + 16 l), so the bank-conflict XOR
//   swizzle is applied to the per-lane SOURCE address and again on the fragment reads (measured SQ_LDS_BANK_CONFLICT = 0)
// * PLANES==2 ("parity" mode): operands are split into bf16 hi + lo and each product is
//   hi*hi + hi*lo + lo*hi, fp32-accumulated -> ~2^-16 relative operand error instead of 2^-9.  hi and lo
//   are stored interleaved per 32-k block ([32 hi | 32 lo] = one 128-byte line, common.h a_pos) so the
//   tile rows are full cache lines in both modes (64-byte half-line DMA requests filled LDS ~1.5x slower)
// * XCD-aware, grouped tile order so that co-resident tiles of one XCD share A panels / W tiles in L2
// * accumulators are kept TRANSPOSED (D^T = W_frag . A_frag^T): a lane then owns 4 consecutive output
//   columns of one row
// * fused epilogues (bias, residual(+row map), exact-erf GELU, bf16 hi/lo split, Q / K / V head scatter), staged through LDS so
//   that every global access is an unconditional 16-byte lane access forming whole row segments (gemm_device.h)
#include <atomic>
#include <map>
#include <mutex>
#include <tuple>
#include <utility>

#include "gemm_device.h"
#include <cstdio>
#include <algorithm>

namespace cwm {

#ifdef CWM_GEMM_PROF
// per-workgroup {s_memrealtime begin, end, s_memtime cycles total, cycles to the end of the main loop} (tools/gemm_prof.py)
__device__ unsigned long long g_gemm_blocks[8192 * 4];
int gemm_prof_dump() {
    static unsigned long long h[8192 * 4];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm_blocks), sizeof(h)) != hipSuccess) return -1;
    FILE* f = fopen("/tmp/gemm_blocks.bin", "wb");
    if (!f) return -1;
    fwrite(h, 1, sizeof(h), f);
    fclose(f);
    return 0;
}
#define GEMM_PROF_BEGIN() const unsigned long long prof_t0 = __builtin_amdgcn_s_memtime(), prof_r0 = __builtin_amdgcn_s_memrealtime(); unsigned long long prof_main = 0
#define GEMM_PROF_MAIN() prof_main = __builtin_amdgcn_s_memtime() - prof_t0
#define GEMM_PROF_END()                                                              \
    do {                                                                             \
        if (threadIdx.x == 0 && blockIdx.x < 8192) {                                 \
            g_gemm_blocks[blockIdx.x * 4 + 0] = prof_r0;                             \
            g_gemm_blocks[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();    \
            g_gemm_blocks[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime() - prof_t0; \
            g_gemm_blocks[blockIdx.x * 4 + 3] = prof_main;                           \
        }                                                                            \
    } while (0)
#else
int gemm_prof_dump() { return -1; }
#define GEMM_PROF_BEGIN() do {} while (0)
#define GEMM_PROF_MAIN() do {} while (0)
#define GEMM_PROF_END() do {} while (0)
#endif

// Tile configurations (BM x BN output tile, WM x WN waves, each wave FM x FN MFMA fragments of 16x16):
//   128x128, 2x2 waves (64x64 per wave)  : 64 KiB LDS, two workgroups per CU         -- small / odd shapes
//   256x128, 4x2 waves (64x64 per wave)  : 96 KiB LDS, one 512-thread workgroup / CU
//   256x256, 2x4 waves (128x64 per wave) : 128 KiB LDS, one 512-thread workgroup / CU -- highest FLOP per staged byte
// The LDS fill (LDS-DMA pieces of 8 rows x 128 B) is the scarce resource (~25 B/clk/CU measured), so bigger
// tiles raise the MFMA ceiling: per K tile a workgroup stages (BM+BN)*128 B and runs BM*BN/256*{2|3} MFMAs.
template <int PLANES, int BM, int BN, int WM, int WN, int STAGES = 2>
__global__ __launch_bounds__(WM * WN * 64, (BM == 128 && BN == 128 && WM * WN == 8 && STAGES == 2) ? 4 : 2) void gemm_bf16_kernel(const GemmParams p) {
    GEMM_PROF_BEGIN();
    constexpr int NWAVES = WM * WN;
    constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
    // every K tile row is one 128-byte line in LDS and in memory: 64 k of the single plane (fast), or
    // 32 k as [32 hi | 32 lo] (parity, interleaved operand layout: common.h a_pos)
    constexpr int BK = 64 / PLANES;           // logical k per tile
    constexpr int NIA = BM / 8 / NWAVES;      // 1-KiB DMA pieces (8 rows each) per wave, A operand
    constexpr int NIB = BN / 8 / NWAVES;      // ... W operand
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static_assert(NIA >= 1 && NIB >= 1, "tile too small for the wave count");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    // split-K workspace: loaded VALUES of the two adjacent pointer fields (hipcc otherwise indexes the kernel-argument struct
    // dynamically and keeps a copy of it in scratch -- the same hazard as qkv_out_base, gemm_device.h)
    float* sk_slabs = p.sk2_slabs;
    unsigned* sk_count = p.sk2_count;
    if constexpr (STAGES > 2) asm volatile("" : "+s"(sk_slabs), "+s"(sk_count));

    // ---- tile selection: XCD chunking + GROUP_M-grouped order --------------------------------
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int ntiles = tiles_m * tiles_n;
    // split-K (deep-ring variants only): blockIdx.x = tile * splitk + part
    const int nsplit = (STAGES > 2 && p.splitk > 1) ? p.splitk : 1;
    const int part = (STAGES > 2) ? (int)blockIdx.x % nsplit : 0;
    const int id = xcd_remap((STAGES > 2) ? (int)blockIdx.x / nsplit : (int)blockIdx.x, ntiles);
    constexpr int GROUP_M = (BM == 128) ? 8 : 4;
    const int group_sz = GROUP_M * tiles_n;
    const int g = id / group_sz;
    const int first_m = g * GROUP_M;
    const int gm = min(tiles_m - first_m, GROUP_M);
    const int in_g = id - g * group_sz;
    const int tile_m = first_m + (in_g % gm);
    const int tile_n = in_g / gm;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-lane source pointers of this wave's DMA pieces (piece j covers tile rows 8j .. 8j+7) ----
    // (32-bit element offsets from the uniform base pointers: one VGPR per piece instead of a 64-bit pointer)
    unsigned a_src[NIA], w_src[NIB];
#pragma unroll
    for (int jj = 0; jj < NIA; ++jj) {
        const int row = (wave * NIA + jj) * 8 + lane / 8;
        const int logical = (lane % 8) ^ lds_swizzle<64>(row);
        a_src[jj] = (unsigned)(p.m_offset + min(m0 + row, p.M - 1)) * (unsigned)(p.lda * PLANES) + logical * 8;
    }
    // direct epilogue with bf16 outputs: the W rows of every 32-row group are staged in permuted order, so that a lane's two column
    // fragments are 8 consecutive output columns (gemm_device.h, epilogue_direct)
    const bool wperm = p.direct && p.epi != EPI_F32;
#pragma unroll
    for (int jj = 0; jj < NIB; ++jj) {
        const int row = (wave * NIB + jj) * 8 + lane / 8;
        const int logical = (lane % 8) ^ lds_swizzle<64>(row);
        w_src[jj] = (unsigned)(n0 + (wperm ? w_row_perm(row) : row)) * (unsigned)(p.K * PLANES) + logical * 8;
    }
    auto issue_tile = [&](int stage, int k0) {
        char* sb = smem + stage * STAGE_BYTES;
        const bf16* ab = p.A + k0;
        const bf16* wb = p.W + k0;
#pragma unroll
        for (int jj = 0; jj < NIA; ++jj)
            __builtin_amdgcn_global_load_lds((gbl_void*)(ab + a_src[jj]), (lds_void*)(sb + (wave * NIA + jj) * 1024), 16, 0, 0);
#pragma unroll
        for (int jj = 0; jj < NIB; ++jj)
            __builtin_amdgcn_global_load_lds((gbl_void*)(wb + w_src[jj]), (lds_void*)(sb + A_BYTES + (wave * NIB + jj) * 1024), 16, 0, 0);
    };
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // 16-byte chunks of a 128-byte row: fast = [k 0..31 | k 32..63] -> two k-steps; parity = [hi | lo] of one k-step
    // The XOR swizzle only involves (row >> 1) & 7 = (frow >> 1) & 7 for every fragment (fragment rows start at
    // multiples of 16), so fragment i sits at a compile-time 2-KiB stride from fragment 0: two base registers per operand.
    const int frow = lane & 15, fq = lane >> 4;
    int a_base[2], b_base[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        a_base[half] = lds_off<64>(wr * (16 * FM) + frow, half * 4 + fq);
        b_base[half] = A_BYTES + lds_off<64>(wc * (16 * FN) + frow, half * 4 + fq);
    }

    const int nk_all = p.K / BK;
    const int kt0 = (STAGES > 2) ? (int)((int64_t)nk_all * part / nsplit) : 0;           // this workgroup's K tiles: [kt0, kt0 + nk)
    const int nk = (STAGES > 2) ? (int)((int64_t)nk_all * (part + 1) / nsplit) - kt0 : nk_all;
    // STAGES == 2: double buffer, the co-resident second workgroup of the CU covers the staging latency.
    // STAGES  > 2 (launches with fewer tiles than CUs, e.g. batch 1: ONE workgroup per CU and nothing to cover for it): a ring of
    // STAGES K tiles with STAGES - 1 tiles of LDS-DMA in flight, counted vmcnt waits and raw barriers; a K tile then costs its MFMA
    // time instead of a full L2 / HBM round trip (ViT-B/8 batch 1, fc2: 88 -> 40 us).
#pragma unroll
    for (int st = 0; st < STAGES - 1; ++st)
        if (st < nk) issue_tile(st, (kt0 + st) * 64);
    for (int t = 0; t < nk; ++t) {
        const int cur = STAGES == 2 ? (t & 1) : t % STAGES;
        if constexpr (STAGES == 2) {
            __syncthreads();  // hipcc drains vmcnt before the barrier: tile t has landed; stage cur^1 is free
            if (t + 1 < nk) issue_tile(cur ^ 1, (t + 1) * 64);  // 64 elements = 128 bytes per tile row
        } else {
            // tiles t .. min(t + STAGES - 2, nk - 1) are in flight; tile t must have landed: allow the younger ones to stay outstanding
            constexpr int PER = NIA + NIB;
            const int younger = min(nk - 1 - t, STAGES - 2);
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();  // every wave's pieces of tile t have landed; every wave is done reading tile t - 1
            __builtin_amdgcn_sched_barrier(0);
            if (t + STAGES - 1 < nk) issue_tile((t + STAGES - 1) % STAGES, (kt0 + t + STAGES - 1) * 64);  // into the stage tile t - 1 occupied
        }
        const char* base = smem + cur * STAGE_BYTES;
        // Software-pipelined fragment reads: a "unit" is one 16-row A fragment (hi[, lo]) against all FN column
        // fragments = FN (fast) / 3 FN (parity) MFMAs.  The A fragments of unit u+DIST are requested before the
        // MFMAs of unit u are issued, so LDS latency hides behind MFMA issue instead of stalling every group.
        constexpr int KSTEPS = BK / 32, UNITS = KSTEPS * FM, DIST = (PLANES == 1) ? 3 : 2;
        bf16x8 bfr[KSTEPS][PLANES][FN];
        bf16x8 afu[UNITS][PLANES];
#pragma unroll
        for (int pl = 0; pl < PLANES; ++pl)
#pragma unroll
            for (int j = 0; j < FN; ++j) bfr[0][pl][j] = *reinterpret_cast<const bf16x8*>(base + b_base[pl] + j * 2048);
#pragma unroll
        for (int u = 0; u < DIST && u < UNITS; ++u)
#pragma unroll
            for (int pl = 0; pl < PLANES; ++pl)
                afu[u][pl] = *reinterpret_cast<const bf16x8*>(base + a_base[PLANES == 1 ? u / FM : pl] + (u % FM) * 2048);
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int kk = u / FM, i = u % FM;
            if (u + DIST < UNITS) {
#pragma unroll
                for (int pl = 0; pl < PLANES; ++pl)
                    afu[u + DIST][pl] =
                        *reinterpret_cast<const bf16x8*>(base + a_base[PLANES == 1 ? (u + DIST) / FM : pl] + ((u + DIST) % FM) * 2048);
            }
            if (PLANES == 1 && KSTEPS == 2 && u == FM - DIST) {  // next k-step's W fragments, ahead of its first unit
#pragma unroll
                for (int j = 0; j < FN; ++j) bfr[KSTEPS - 1][0][j] = *reinterpret_cast<const bf16x8*>(base + b_base[1] + j * 2048);
            }
            // transposed accumulation D^T = W_frag . A_frag^T: the lane owns 4 consecutive output columns of one row
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if constexpr (PLANES == 2) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][0][j], afu[u][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][1][j], afu[u][0], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][0][j], afu[u][0], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (p.debug & 2) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    GEMM_PROF_MAIN();
    static_assert(STAGES <= 4, "the counted waits above cover at most 3 tiles in flight");
    if constexpr (STAGES > 2) {
        if (nsplit > 1) {
            // ---- split-K: every part writes its fp32 accumulators to its slab; the part that arrives LAST adds all slabs of the tile
            // in part order (its own included, read back: the sum does not depend on who was last -> deterministic) and runs the epilogue.
            // Hand-off (MI355X guide, Guideline 16): plain stores -> every storing wave's vmcnt(0) -> workgroup barrier -> one lane:
            // agent-scope release, returning agent-scope atomic add; last part: agent-scope acquire, vmcnt(0), barrier, plain loads.
            const int tile_id = (int)blockIdx.x / nsplit;
            float* slab = sk_slabs + ((size_t)tile_id * nsplit + part) * (BM * BN) + (size_t)wave * (FM * FN * 256) + lane * 4;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) *reinterpret_cast<f32x4*>(slab + (i * FN + j) * 256) = acc[i][j];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int* flag = reinterpret_cast<int*>(smem);  // (the operand ring is dead: every wave passed the barrier after its last MFMA)
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned prev = __hip_atomic_fetch_add(sk_count + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool last = prev == (unsigned)(nsplit - 1);
                if (last) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(sk_count + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
                }
                *flag = last ? 1 : 0;
            }
            __syncthreads();
            if (*flag == 0) return;
            const float* base_slab = sk_slabs + (size_t)tile_id * nsplit * (BM * BN) + (size_t)wave * (FM * FN * 256) + lane * 4;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = *reinterpret_cast<const f32x4*>(base_slab + (i * FN + j) * 256);
            for (int q = 1; q < nsplit; ++q) {
                const float* sl = base_slab + (size_t)q * (BM * BN);
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] += *reinterpret_cast<const f32x4*>(sl + (i * FN + j) * 256);
            }
        }
    }
    if (p.direct) {
        __syncthreads();  // every wave is done with the operand tiles: the row table takes their place
        int4* tab = reinterpret_cast<int4*>(smem);
        epilogue_row_table(p, tab, m0, BM, tid);
        __syncthreads();
        epilogue_direct_tile<PLANES, FM, FN>(p, acc, tab, wr * (16 * FM), n0 + wc * (16 * FN), lane);
    } else if (p.staged) {
        __syncthreads();  // every wave is done with the operand tiles: LDS becomes the epilogue's staging space
        int4* tab = reinterpret_cast<int4*>(smem + NWAVES * 8192);
        epilogue_row_table(p, tab, m0, BM, tid);
        __syncthreads();
        epilogue_staged<PLANES, FM, FN>(p, acc, smem + wave * 8192, tab, wr * (16 * FM), n0 + wc * (16 * FN), lane);
    } else {
        epilogue_rows<PLANES, FM, FN>(p, acc, m0, n0, wr, wc, lane);
    }
    GEMM_PROF_END();
}

// ---------------------------------------------------------------------------------------------------
// 256x256 "8-phase" kernel: the deep-pipelined structure for wide outputs (qkv, fc1).
//
// The kernel above waits vmcnt(0) at one workgroup barrier per K tile with all 8 waves in lockstep, so every
// tile starts with a dead LDS-read burst and ends by draining its LDS-DMA: MFMA busy stalls near 40 %.  Here
//  * the 256x256 tile is cut in four 128x128 quadrants; in every quadrant wave (wr, wc) owns a 64x32 piece,
//    and a K tile is cut in four 16-KiB half-tiles (A rows 0-127 | A rows 128-255 | W rows 0-127 | W rows 128-255)
//  * a K tile is 4 phases, one quadrant each: { ds_read the half-tile fragments the quadrant adds, issue the
//    LDS-DMA of ONE half-tile (2 pieces per wave) -> s_barrier -> MFMA cluster -> s_barrier }
//  * the two wave groups (wr = 0 / 1, one wave of each per SIMD) run one barrier apart, so on every SIMD one
//    wave's MFMA cluster covers the other wave's load segment
//  * LDS-DMA stays in flight ACROSS the barriers: raw s_barrier, and a counted s_waitcnt vmcnt(6) once per K
//    tile (phase 4) that leaves the three youngest half-tiles in flight.
// Half-tile schedule (t = K tile, buffer t & 1):   P1(t): A1(t+1)   P2(t): A0(t+2)   P3(t): W0(t+2)   P4(t): W1(t+2)
// so the wait in P4(t) retires all of tile t+1, which is first read one barrier later, in P1(t+1).
// Half-width column tile (round 5): when the tile's columns 128 .. 255 all lie past N (N mod 256 in 1 .. 128: the decoder's N = 384 and 1152), the
// quadrants (., 1) are all padding -- their MFMA clusters, their W1 fragment reads and the W1 half-tile's LDS-DMA are skipped (every barrier stays:
// the hand-off reasoning below is unchanged), the counted wait becomes vmcnt(4) (a K tile issues three half-tiles instead of four) and the epilogue
// runs over two pieces instead of four.  A half tile holds a CU for ~0.6 of a full tile's time; with it N = 384 runs as 1.5 column tiles on this
// kernel instead of three 128x128 tiles per row block on the L2-feed-bound small kernel.  Same product sequence per accumulator: bit-identical.
// Write-after-read: A half-tiles are staged by the group that reads them (waves 0-3 stage and read rows 0-63,
// waves 4-7 rows 64-127), so one phase after the reads suffices; W half-tiles are read by both groups, and
// the leading group re-stages them two phases after the read (W0 read in P1 -> staged in P3, W1 P2 -> P4).

// One 256x256 tile (HALF: its columns 128 .. 255 are all padding -- see above) of the 8-phase kernel.
template <int PLANES, bool HALF>
__device__ __forceinline__ void gemm8p_tile(const GemmParams& p, char* smem, int m0, int n0) {
    GEMM_PROF_BEGIN();
    constexpr int HALF_BYTES = 128 * 128;      // 128 rows x 128 B
    constexpr int BUF_BYTES = 4 * HALF_BYTES;  // A0 | A1 | W0 | W1
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- LDS-DMA sources: wave w stages pieces 2w, 2w+1 (rows 16w .. 16w+15) of every half-tile ----
    // (direct epilogue with bf16 outputs: W rows permuted inside every 32-row group, gemm_device.h epilogue_direct)
    const bool wperm = p.direct && p.epi != EPI_F32;
    unsigned src[4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int row = wave * 16 + jj * 8 + lane / 8;
            const int logical = (lane % 8) ^ lds_swizzle<64>(row);
            src[h][jj] = (unsigned)(p.m_offset + min(m0 + h * 128 + row, p.M - 1)) * (unsigned)(p.lda * PLANES) + logical * 8;
            src[2 + h][jj] = (unsigned)(n0 + h * 128 + (wperm ? w_row_perm(row) : row)) * (unsigned)(p.K * PLANES) + logical * 8;
        }
    auto stage = [&](auto half_c, int buf, int kt) {
        constexpr int half = decltype(half_c)::value;
        const bf16* gb = (half < 2 ? p.A : p.W) + (size_t)kt * 64;  // 64 elements = 128 bytes per tile row
        char* sb = smem + buf * BUF_BYTES + half * HALF_BYTES + wave * 2048;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
            __builtin_amdgcn_global_load_lds((gbl_void*)(gb + src[half][jj]), (lds_void*)(sb + jj * 1024), 16, 0, 0);
    };
    using H_A0 = std::integral_constant<int, 0>;
    using H_A1 = std::integral_constant<int, 1>;
    using H_W0 = std::integral_constant<int, 2>;
    using H_W1 = std::integral_constant<int, 3>;

    // ---- fragment addresses (XOR swizzle only involves (frow >> 1) & 7: fragments sit at 2-KiB strides) ----
    const int frow = lane & 15, fq = lane >> 4;
    int a_base[2], b_base[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        a_base[h] = lds_off<64>(wr * 64 + frow, h * 4 + fq);
        b_base[h] = 2 * HALF_BYTES + lds_off<64>(wc * 32 + frow, h * 4 + fq);
    }
    bf16x8 af[2][4], bw0[2][2], bw1[2][2];  // [16-byte chunk set: k-step (fast) | hi, lo (parity)][fragment]
    auto read_a = [&](const char* base, int qm) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) af[h][i] = *reinterpret_cast<const bf16x8*>(base + qm * HALF_BYTES + a_base[h] + i * 2048);
    };
    auto read_w = [&](const char* base, int qn, bf16x8 (&bw)[2][2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) bw[h][j] = *reinterpret_cast<const bf16x8*>(base + qn * HALF_BYTES + b_base[h] + j * 2048);
    };

    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed accumulation D^T = W_frag . A_frag^T (the lane owns 4 consecutive output columns of one row)
    auto mfma_quadrant = [&](f32x4 (&c)[4][2], const bf16x8 (&bw)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
        if constexpr (PLANES == 1) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[h][j], af[h][i], c[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[0][j], af[1][i], c[i][j], 0, 0, 0);
                    c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[1][j], af[0][i], c[i][j], 0, 0, 0);
                    c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[0][j], af[0][i], c[i][j], 0, 0, 0);
                }
        }
        __builtin_amdgcn_s_setprio(0);
    };
#define CWM_PHASE_BARRIER()                   \
    do {                                      \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
    } while (0)

    const int nk = p.K / (64 / PLANES);
    // ---- prologue: all of tile 0, then A0 / W0 / W1 of tile 1 ----
    // (the waits leave tile 1's pieces in flight: three half-tiles = 6 pieces, two = 4 for a half-width tile)
    if constexpr (!HALF) {
        stage(H_A0{}, 0, 0);
        stage(H_W0{}, 0, 0);
        stage(H_W1{}, 0, 0);
        stage(H_A1{}, 0, 0);
        if (nk > 1) {
            stage(H_A0{}, 1, 1);
            stage(H_W0{}, 1, 1);
            stage(H_W1{}, 1, 1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        stage(H_A0{}, 0, 0);
        stage(H_W0{}, 0, 0);
        stage(H_A1{}, 0, 0);
        if (nk > 1) {
            stage(H_A0{}, 1, 1);
            stage(H_W0{}, 1, 1);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    CWM_PHASE_BARRIER();
    if (wr == 1) CWM_PHASE_BARRIER();  // the second wave group runs one barrier behind the first

    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        const char* base = smem + buf * BUF_BYTES;
        // ---- P1: quadrant (0, 0) ----
        read_w(base, 0, bw0);
        read_a(base, 0);
        if (t + 1 < nk) stage(H_A1{}, buf ^ 1, t + 1);
        CWM_PHASE_BARRIER();
        mfma_quadrant(acc[0][0], bw0);
        CWM_PHASE_BARRIER();
        // ---- P2: quadrant (0, 1) ----
        if constexpr (!HALF) read_w(base, 1, bw1);
        if (t + 2 < nk) stage(H_A0{}, buf, t + 2);
        CWM_PHASE_BARRIER();
        if constexpr (!HALF) mfma_quadrant(acc[0][1], bw1);
        CWM_PHASE_BARRIER();
        // ---- P3: quadrant (1, 1) ----
        read_a(base, 1);
        if (t + 2 < nk) stage(H_W0{}, buf, t + 2);
        CWM_PHASE_BARRIER();
        if constexpr (!HALF) mfma_quadrant(acc[1][1], bw1);
        CWM_PHASE_BARRIER();
        // ---- P4: quadrant (1, 0); retire tile t+1 (A1(t+1) was issued in P1: the two / three half-tiles issued since may stay in flight) ----
        if (t + 2 >= nk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if constexpr (HALF) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            stage(H_W1{}, buf, t + 2);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
        CWM_PHASE_BARRIER();
        mfma_quadrant(acc[1][0], bw0);
        CWM_PHASE_BARRIER();
    }
    if (wr == 0) CWM_PHASE_BARRIER();  // balance the stagger barrier
#undef CWM_PHASE_BARRIER

    if (p.debug & 2) {
#pragma unroll
        for (int qm = 0; qm < 2; ++qm)
#pragma unroll
            for (int qn = 0; qn < 2; ++qn)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[qm][qn][i][j]));
        return;
    }
    GEMM_PROF_MAIN();
    // (the balancing barrier above is also the point where every wave is done reading the operand tiles)
    if (p.direct) {
        // (Building this table in the kernel's prologue, behind the operand ring, and dropping the barrier pair that only protects the ring --
        // so that every wave goes from its last MFMA straight to its stores -- measured equal on every GEMM shape and 0.7-1 % SLOWER on the
        // step, profiles/r4_ab_gemm_prologue_table.log: the epilogue is bound by the output stores, not by its start-up.)
        int4* tab = reinterpret_cast<int4*>(smem);
        epilogue_row_table(p, tab, m0, 256, tid);
        __syncthreads();
        if constexpr (HALF)
            epilogue_direct<PLANES, 2>(
                p, [&](int pi, int i, int j) { return acc[pi][0][i][j]; }, [&](int pi) { return pi * 128 + wr * 64; }, [&](int) { return n0 + wc * 32; }, tab, lane);
        else
            epilogue_direct<PLANES, 4>(
                p, [&](int pi, int i, int j) { return acc[pi >> 1][pi & 1][i][j]; }, [&](int pi) { return (pi >> 1) * 128 + wr * 64; },
                [&](int pi) { return n0 + (pi & 1) * 128 + wc * 32; }, tab, lane);
        GEMM_PROF_END();
        return;
    }
    if (p.staged) {
        int4* tab = reinterpret_cast<int4*>(smem + 8 * 8192);
        epilogue_row_table(p, tab, m0, 256, tid);
        __syncthreads();
        if constexpr (HALF)
            epilogue_piece_seq<PLANES, 2>(
                p, [&](int pi, int i, int j) { return acc[pi][0][i][j]; }, [&](int pi) { return pi * 128 + wr * 64; }, [&](int) { return n0 + wc * 32; },
                smem + wave * 8192, tab, lane);
        else
            epilogue_piece_seq<PLANES, 4>(
                p, [&](int pi, int i, int j) { return acc[pi >> 1][pi & 1][i][j]; }, [&](int pi) { return (pi >> 1) * 128 + wr * 64; },
                [&](int pi) { return n0 + (pi & 1) * 128 + wc * 32; }, smem + wave * 8192, tab, lane);
        GEMM_PROF_END();
        return;
    }
    const int ncol = (lane >> 4) * 4;
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + qm * 128 + wr * 64 + i * 16 + (lane & 15);
            if (m >= p.M) continue;
            const RowMap rm = map_row(p, m + p.m_offset);
#pragma unroll
            for (int qn = 0; qn < 2; ++qn)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int nb = n0 + qn * 128 + wc * 32 + j * 16;
                    if (nb >= p.N) continue;
                    epilogue_frag<PLANES>(p, rm, nb, ncol, acc[qm][qn][i][j]);
                }
        }
}

template <int PLANES>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_m = (p.M + 255) / 256;
    const int tiles_n = (p.N + 255) / 256;
    const int id = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    constexpr int GROUP_M = 4;  // (group heights 1 .. 12 measured within +-1 % of each other on every model shape)
    const int group_sz = GROUP_M * tiles_n;
    const int g = id / group_sz;
    const int first_m = g * GROUP_M;
    const int gm = min(tiles_m - first_m, GROUP_M);
    const int in_g = id - g * group_sz;
    const int m0 = (first_m + (in_g % gm)) * 256, n0 = (in_g / gm) * 256;
    // two instances of the tile, chosen once per workgroup (launch-uniform per tile column): each has its own main loop with ONE counted wait --
    // no branch on the tile kind inside the loops (tools/asm_lds_lint.py checks the two loops against their own piece counts)
    if (n0 + 128 >= p.N && !(p.debug & 1024)) gemm8p_tile<PLANES, true>(p, smem, m0, n0);
    else gemm8p_tile<PLANES, false>(p, smem, m0, n0);
}

// Split-K workspace of the deep-ring kernel: kSplitKSlots fp32 slabs of one 128x128 tile + arrival counters (zeroed once; the kernel
// leaves them at zero).  One workspace serves ONE stream at a time: the engine keeps one per stream it launches on (engine.hip).
int splitk_workspace_alloc(float** slabs, unsigned** counts, hipStream_t stream) {
    CWM_HIP_CHECK(hipMalloc((void**)slabs, (size_t)kSplitKSlots * 128 * 128 * sizeof(float)));
    CWM_HIP_CHECK(hipMalloc((void**)counts, kSplitKSlots * sizeof(unsigned)));
    // zeroed ON THE STREAM that will use the counters: ordered before the first launch there without a device-wide synchronisation
    // (the kernels leave the counters at zero, so this runs once per workspace)
    CWM_HIP_CHECK(hipMemsetAsync(*counts, 0, kSplitKSlots * sizeof(unsigned), stream));
    return 0;
}

// Does this launch take the split-K path of the deep-ring kernel?  (the engine creates a stream's workspace only then)
// Tile height of the deep-ring kernel for a launch with at most one 128x128 tile per CU: 64 rows when the 128-row grid would leave half of the
// CUs idle (batch 1: 792 rows = 7 row tiles, the last one 24 rows; qkv 126 tiles on 256 CUs -> 13 x 18 = 234 tiles of half the work each).
// "gemm_debug" bit 8 keeps 128.
static inline const Tuning& tn(const GemmParams& p) { return p.tune ? *p.tune : default_tuning(); }

static int deep_tile_rows(const GemmParams& p) {
    const int tiles128 = ((p.M + 127) / 128) * ((p.N + 127) / 128);
    return (tiles128 * 2 <= gemm_cu_count() && p.M > 64 && !(tn(p).gemm_debug & 256)) ? 64 : 128;
}

int gemm_splitk_parts(const GemmParams& p, int planes) {
    if (tn(p).gemm_debug & (4 | 32)) return 1;
    const int tiles128 = ((p.M + 127) / 128) * ((p.N + 127) / 128);
    const int cus = gemm_cu_count();
    if (tiles128 > cus) return 1;
    const int bm = deep_tile_rows(p);
    const int tiles = ((p.M + bm - 1) / bm) * ((p.N + 127) / 128);
    const int nk_all = p.K / (64 / planes);
    const int sk = std::min(std::min(cus / tiles, nk_all / 12), 8);
    return sk >= 3 ? sk : 1;
}

int gemm_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        cus = prop.multiProcessorCount & ~7;
        if (cus < 8) cus = 256;
    }
    return cus;
}

static int launch_gemm_cfg(GemmParams& p, int planes, int cfg, hipStream_t stream);

// Mixed tiling (tile configuration 6): the leading rows that fill whole rounds of 256x256 tiles go to the 8-phase kernel, the
// remaining rows to 128x128 tiles.  Both kernels apply the same product sequence to every accumulator, so the result does not
// depend on the split.  Returns false if the shape has no such split.
bool gemm_mixed_split(const GemmParams& p, GemmParams* big, GemmParams* rest) {
    const int tiles_n = (p.N + 255) / 256, tiles_m = (p.M + 255) / 256;
    const int cus = gemm_cu_count();
    const int rounds = (tiles_m * tiles_n) / cus;
    const int big_rows = std::min(tiles_m - 1, rounds * cus / tiles_n);  // m-tile rows of the 8-phase part
    if (rounds < 1 || big_rows < 1) return false;
    *big = p;
    *rest = p;
    big->M = big_rows * 256;
    rest->m_offset = p.m_offset + big->M;
    rest->M = p.M - big->M;
    return true;
}

static int launch_gemm_checked(const GemmParams& p_in, int planes, int forced_cfg, hipStream_t stream);

// One launch with a given tile configuration (1 .. 5), after launch_gemm's argument checks.  (The configuration is an explicit
// argument all the way down: no process-global is touched, so forwards on two host threads cannot race on it.)
int launch_gemm_tile(const GemmParams& p_in, int planes, int cfg, hipStream_t stream) { return launch_gemm_checked(p_in, planes, cfg, stream); }

int launch_gemm(const GemmParams& p_in, int planes, hipStream_t stream) { return launch_gemm_checked(p_in, planes, 0, stream); }

// Tile configuration for a launch: Tuning.gemm_tile (development switch) if set, else the development library's per-shape hook, else the measured rule.
int gemm_choose_tile(const GemmParams& p, int planes) {
    (void)planes;
    const Tuning& t = tn(p);
    int cfg = t.gemm_tile;
    if (cfg == 0 && t.tile_hook) cfg = t.tile_hook(p.M, p.N, p.K, p.epi, p.overlapped ? 1 : 0);
    // 0 auto, 1: 128x128, 4: 256x256 8-phase, 6: 4 + 1 by rows
    if (cfg == 0) {
        // Measured on MI355X (tools/microbench.py gemm / gemm_mid / gemm_l4: B/8 at batch 8, 16, 32 and L/4 batch 8, both modes;
        // profiles/r1n_*, r1o_*, r1p_* logs).  The 256x256 8-phase kernel has the fastest main loop (~1.6 PFLOP/s of executed MFMA
        // work) but one workgroup per CU, so what decides is how its grid fills the chip:
        //  * fewer tiles than CUs: one partial round -- fine from half the CUs up, else 128x128 tiles (more, smaller workgroups)
        //  * whole rounds + a last round at least half full: 8-phase
        //  * otherwise mixed tiling by rows (6): whole rounds of 256x256 tiles + a remainder of 128x128 tiles -- except the small
        //    short-K launches (proj of B/8), which stay on 128x128 tiles with two workgroups per CU
        //  * N < 512 or ragged narrow N (N = 384, head, patch embed): 128x128
        // (A persistent stream-K form of the 8-phase kernel existed in rounds 1-3, measured no faster than (6), never selected and removed
        // in round 4 -- git history)
        //  * K < 512 (the B/8 decoder: K = 384), fp32 outputs: the 8-wave 128x128 kernel (its epilogues overlap the co-resident workgroup's
        //    main loop, and with 12 K tiles the epilogue is a third of a 256x256 tile's time).  bf16 outputs (qkv, fc1) since round 4: with
        //    the direct epilogue the 8-phase kernel wins there too (decoder qkv 149 -> 140 us, fc1 214 -> 191 us, profiles/r4_ab_gemm_direct.log)
        cfg = 1;
        const bool bf16_out = p.epi != EPI_F32;
        // N: whole 256-column tiles, or wide enough that a ragged last one weighs little -- or, since round 5, a last tile that is exactly a half-width
        // one (N mod 256 = 128: the decoder's N = 384 as 1.5 column tiles; "gemm_debug" bit 1024 restores the round-4 rule and kernel)
        const bool half_ok = !(t.gemm_debug & 1024) && p.N >= 384 && p.N % 256 == 128;
        if (p.K >= ((bf16_out && !(t.gemm_debug & 512)) ? 256 : 512) && p.M >= 512 && (p.N >= 1024 || (p.N >= 512 && p.N % 256 == 0) || half_ok)) {
            // (a half-width last column tile holds its CU for ~0.6 of a full tile's time: counted as such when the fill of the rounds is judged)
            const int64_t tiles_m = (p.M + 255) / 256, tiles_n = (p.N + 255) / 256;
            const int64_t tiles = (half_ok && p.N % 256 == 128) ? tiles_m * (tiles_n - 1) + (tiles_m * 3 + 4) / 5 : tiles_m * tiles_n;
            const int cus = gemm_cu_count();
            if (tiles < cus) {
                // one partial round: from half the CUs up.  Inside a two-lane call the other lane fills the idle CUs, and the launches whose tile is long (K >= 1024:
                // fc2) or whose epilogue is the direct one (bf16 outputs) win on the 8-phase kernel from 0.4 of the CUs (tools/autotune_step.py and
                // tools/batch_sweep.py, round 4: ViT-B/8 batch 24 -- fc2 = 114 tiles -- 13.03 -> 12.53 ms per step, batch 8 -- qkv = 117 tiles -- 5.12 -> 5.09;
                // 75 and 93 tiles (batch 16, 20) measure equal or worse, 57 and 39 lose)
                const bool long_or_direct = bf16_out || p.K >= 1024;
                cfg = (tiles * 2 >= cus || (p.overlapped && long_or_direct && tiles * 5 >= cus * 2)) ? 4 : 1;
            } else if (p.overlapped && !(t.gemm_debug & 128)) {
                // two batch lanes: the other lane's kernels take the CUs a partly filled last round leaves idle, so the kernel with the
                // fastest main loop wins regardless of the fill (B/8 batch 32 as 2 x 16: +2 % over the single-lane rule below)
                cfg = 4;
            } else {
                // whole rounds on the 8-phase kernel; a last round that is less than half full (80 % for the short-K fp32 launches, where the
                // un-overlapped epilogue weighs more) costs a whole tile period there, and its rows are cheaper on 128x128 tiles (mixed tiling,
                // ~0.8 of the 8-phase rate): qkv at batch 32 = 891 tiles = 3.48 rounds, 254 -> 228 us (profiles/r3c_ab_mfma_order.log).  Round 4
                // re-measured the threshold with the direct epilogue (profiles/r4_ab_gemm_direct.log): 48 % full -> mixed (222 vs 228 us),
                // 55-64 % full -> 8-phase (ViT-L/4 qkv 351 vs 372, fc2 486 vs 507, decoder fc2 275 vs 291).  The small short-K launches
                // (proj: K, N < 1024) stay on 128x128 tiles.
                const int last = (int)(tiles % cus);
                const bool full_enough = (bf16_out || p.K >= 1024) ? last * 2 >= cus : last * 5 >= cus * 4;
                if (last == 0 || full_enough) cfg = 4;
                else if (half_ok && p.N < 512) cfg = tiles < 2 * cus ? 1 : 4;  // N = 384 alone on the chip: a badly filled SECOND round goes to the 128x128 kernel
                                                                               // (dec.fc2 of a one-lane batch 32: 195 us either way in parity mode, 78 vs 88 us in
                                                                               // fast mode), from two whole rounds on the 8-phase kernel wins (the IMU model's
                                                                               // batch-16 dec.fc2: 373 -> 358 us; profiles/r5_mb_halftile.log)
                else if (p.K >= 1024 || p.N >= 1024) cfg = 6;
            }
        }
    }
    return cfg;
}

static int launch_gemm_checked(const GemmParams& p_in, int planes, int forced_cfg, hipStream_t stream) {
    GemmParams p = p_in;
    const Tuning& t = tn(p);
    p.debug = t.gemm_debug;
    CWM_REQUIRE(planes == 1 || planes == 2, "gemm: planes must be 1 or 2");
    CWM_REQUIRE(p.K % 64 == 0, "gemm: K=%d must be a multiple of 64", p.K);
    CWM_REQUIRE(p.lda % 8 == 0, "gemm: lda=%d must be a multiple of 8", p.lda);
    CWM_REQUIRE(p.M > 0 && p.N > 0, "gemm: empty problem M=%d N=%d", p.M, p.N);
    CWM_REQUIRE(p.N % 16 == 0, "gemm: N=%d must be a multiple of 16", p.N);
    if (p.epi == EPI_F32) {
        CWM_REQUIRE(p.ldc % 4 == 0 && (!p.resid || p.ldr % 4 == 0), "gemm: ldc/ldr must be multiples of 4");
    } else if (p.epi == EPI_QKV) {
        CWM_REQUIRE(p.rows_in == p.n_tok && p.N == 3 * p.qkv_dim && p.head_dim % 4 == 0, "gemm: bad QKV epilogue setup");
        CWM_REQUIRE(p.qkv_dim % 16 == 0, "gemm: QKV epilogue needs the model width (%d) to be a multiple of 16", p.qkv_dim);
    } else {
        CWM_REQUIRE(p.ldo % 4 == 0, "gemm: ldo must be a multiple of 4");
        CWM_REQUIRE(planes == 1 || p.ldo % 32 == 0, "gemm: split-bf16 output rows are whole [32 hi | 32 lo] blocks: ldo=%d must be a multiple of 32", p.ldo);
    }
    // ---- LDS-staged epilogue whenever its 16-byte row segments are aligned (always, for the predictor's widths) ----
    p.staged = 0;
    if (t.gemm_staged) {
        if (p.epi == EPI_F32) p.staged = 1;
        else if (p.epi == EPI_QKV) p.staged = (p.qkv_dim % 32 == 0 && p.head_dim % 32 == 0);
        else p.staged = (p.ldo % 8 == 0);
    }
    // LDS-DMA source addresses are 32-bit element offsets from the operand base pointers (one VGPR per piece): the operands must stay
    // below 2^32 elements (parity-mode fc2 of ViT-L/4 reaches that at batch ~165: chunk M on the host beyond it)
    CWM_REQUIRE((int64_t)(p.m_offset + p.M) * p.lda * planes < (1ll << 32) && (int64_t)(((p.N + 255) / 256) * 256) * p.K * planes < (1ll << 32),
                "gemm: operand too large for 32-bit element offsets (M=%d lda=%d N=%d K=%d planes=%d): split the batch", p.m_offset + p.M, p.lda,
                p.N, p.K, planes);
    p.direct = (t.gemm_direct && p.staged && (p.epi != EPI_F32 || t.gemm_direct >= 2)) ? 1 : 0;
    int cfg = forced_cfg > 0 ? forced_cfg : gemm_choose_tile(p, planes);
    if (cfg == 6) {
        GemmParams a, b;
        if (gemm_mixed_split(p, &a, &b)) {
            if (int rc = launch_gemm_cfg(a, planes, 4, stream)) return rc;
            return launch_gemm_cfg(b, planes, 1, stream);
        }
        cfg = 1;
    }
    return launch_gemm_cfg(p, planes, cfg, stream);
}

static int launch_gemm_cfg(GemmParams& p, int planes, int cfg, hipStream_t stream) {
    typedef void (*kern_t)(const GemmParams);
    CWM_REQUIRE(cfg == 1 || cfg == 4, "gemm: unknown tile configuration %d (1: 128x128, 4: 256x256 8-phase)", cfg);
    const int dbg = tn(p).gemm_debug;
    if (cfg == 4) {
        static const kern_t k8[2] = {gemm8p_kernel<1>, gemm8p_kernel<2>};
        const size_t smem8 = 2 * 4 * 128 * 128;
        kern_t k = k8[planes - 1];
        if (int rc = cwm_set_max_lds((const void*)k, (int)smem8)) return rc;
        const int tiles8 = ((p.M + 255) / 256) * ((p.N + 255) / 256);
        hipLaunchKernelGGL(k, dim3(tiles8), dim3(512), smem8, stream, p);
        CWM_HIP_CHECK(hipGetLastError());
        return 0;
    }
    // 128x128 tiles, at most one workgroup per CU (fewer tiles than CUs): the 4-stage ring hides the staging latency that the second
    // co-resident workgroup hides in bigger launches ("gemm_debug" bit 2 switches it off for A/B runs)
    const int tiles128 = ((p.M + 127) / 128) * ((p.N + 127) / 128);
    const int cus = gemm_cu_count();
    const bool deep = !(dbg & 4) && tiles128 <= cus;
    p.splitk = 1;
    if (deep && !(dbg & 32)) {
        // fill the idle CUs of a latency-bound launch by cutting K: only where it pays (measured, ViT-B/8 batch 1: fc2 65 -> 32 us with
        // 6 parts, decoder fc2 36 -> 26 us; two parts of a K = 768 qkv projection LOSE 6 us to the hand-off) -- at least three parts of at
        // least 12 K tiles each; the parts of a tile are reduced in a fixed order (deterministic)
        const int sk = gemm_splitk_parts(p, planes);
        if (sk >= 3) {
            if (!p.sk2_slabs) {
                // callers without a workspace of their own (the stand-alone entry points, cwm_linear): one workspace per (device,
                // stream), created under a lock -- launches on one stream are ordered, so they may share it; two streams never do
                struct Ws { float* slabs; unsigned* counts; };
                static std::mutex mu;
                static std::map<std::pair<int, hipStream_t>, Ws> table;
                int dev = 0;
                CWM_HIP_CHECK(hipGetDevice(&dev));
                std::lock_guard<std::mutex> lock(mu);
                auto it = table.find(std::make_pair(dev, stream));
                if (it == table.end()) {
                    Ws w = {nullptr, nullptr};
                    if (int rc = splitk_workspace_alloc(&w.slabs, &w.counts, stream)) return rc;
                    it = table.emplace(std::make_pair(dev, stream), w).first;
                }
                p.sk2_slabs = it->second.slabs;
                p.sk2_count = it->second.counts;
            }
            p.splitk = sk;
        }
    }
    if (deep) {
        // 8 waves of 64x32 (two per SIMD cover each other's LDS / barrier latency; the 4-wave form measured slower and was removed in round 5), or
        // 64x128 tiles as 4 waves of 64x32 where the 128-row grid would leave half of the CUs idle
        static const kern_t deep128w8[2] = {gemm_bf16_kernel<1, 128, 128, 2, 4, 4>, gemm_bf16_kernel<2, 128, 128, 2, 4, 4>};
        static const kern_t deep64[2] = {gemm_bf16_kernel<1, 64, 128, 1, 4, 4>, gemm_bf16_kernel<2, 64, 128, 1, 4, 4>};
        const int bm = deep_tile_rows(p);
        const size_t smem = (size_t)4 * (bm + 128) * 128;
        kern_t k = bm == 64 ? deep64[planes - 1] : deep128w8[planes - 1];
        const int tiles = ((p.M + bm - 1) / bm) * ((p.N + 127) / 128);
        if (int rc = cwm_set_max_lds((const void*)k, (int)smem)) return rc;
        hipLaunchKernelGGL(k, dim3(tiles * p.splitk), dim3(bm == 64 ? 256 : 512), smem, stream, p);
        CWM_HIP_CHECK(hipGetLastError());
        return 0;
    }
    // 128x128 tiles as 8-wave workgroups (64x32 per wave), two per CU = FOUR waves per SIMD: measured 5-25 % faster than the 4-wave form (two
    // waves per SIMD) on every model shape -- the extra waves cover the per-K-tile barrier + LDS-DMA latency that the simple loop exposes.
    // (Rounds 1-4 also carried 256x128 and 256x256 tiles on this loop and the 4-wave forms: never selected by the rule above, removed in round 5.)
    static const kern_t kerns[2] = {gemm_bf16_kernel<1, 128, 128, 2, 4>, gemm_bf16_kernel<2, 128, 128, 2, 4>};
    // (the epilogue's eight 8-KiB wave buffers fill the 64-KiB operand ring; the row table sits behind them)
    const size_t smem = (size_t)2 * (128 + 128) * 128 + 4096;
    const int tiles = ((p.M + 127) / 128) * ((p.N + 127) / 128);
    kern_t k = kerns[planes - 1];
    CWM_REQUIRE(smem >= (size_t)8 * 8192 + (size_t)128 * 16 + (size_t)128 * 8, "gemm: dynamic LDS too small for the staged epilogue (wave buffers + row table)");
    if (int rc = cwm_set_max_lds((const void*)k, (int)smem)) return rc;
    hipLaunchKernelGGL(k, dim3(tiles), dim3(512), smem, stream, p);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace cwm
