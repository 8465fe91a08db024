// Persistent stream-K form of the 256x256 8-phase GEMM (gemm.hip) for CDNA4:  C[M,N] = A[M,K] * W[N,K]^T  (+ epilogue)
//
// Why: with one 512-thread workgroup per CU and K = 384 .. 3072, the launch-per-tile kernel loses
//   * a wave-quantisation tail (891 tiles on 256 CUs run as 4 rounds for 3.5 rounds of work; N = 768 outputs give
//     297 tiles = 2 rounds for 1.2, so they could not use the 256x256 tile at all),
//   * a cold 7-half-tile prologue per tile, and
//   * lock-step epilogues: every CU writes its 256-KiB tile at the same moment, an HBM-rate burst (measured 40 us of a
//     240 us QKV projection) that nothing overlaps.
// Here the grid is one workgroup per CU (G workgroups, T tiles).  The first S = G + T mod G tiles (the "stream-K region")
// are cut, as (tile, k-tile) units, into G equal contiguous spans of 1 .. 2 tiles; the other T - S tiles are dealt out
// whole, one per workgroup and round (workgroups that share an XCD take neighbouring tiles, so A panels / W tiles are
// shared in L2 exactly as in the launch-per-tile kernel -- a pure stream-K split of ALL tiles put the concurrently
// running workgroups 3-4 tiles apart and lost that: main loop 20 % slower at K = 3072).  A workgroup walks its units with
// the 8-phase main loop running THROUGH tile boundaries (the LDS ring always holds the next two k-tiles, whatever tile
// they belong to) and runs a tile's epilogue when its last k-tile is done.  Spans end at different k offsets, so from then
// on the epilogues of different CUs are spread over the whole launch, and every workgroup does exactly T / G tiles of work.
//
// Split tiles.  A span boundary inside a tile leaves its head k-range to workgroup w and its tail to w + 1 (a span is
// never shorter than one tile, so there are at most two owners).  A workgroup walks its tiles in DESCENDING order (k
// ascending inside a tile): it first meets its head part, whose fp32 accumulators it writes to slab[w] and publishes
// (flag[w] = epoch), and it ends with its tail part, for which it waits for flag[w - 1], adds slab[w - 1] and runs the
// epilogue.  The producer therefore always has the smaller workgroup index and publishes before anything it could wait
// for: with in-order dispatch the wait cannot deadlock whatever the number of resident workgroups (the spin is bounded
// and reports through p.sk_err regardless).  Hand-off = MI355X guide, Guideline 16 form R1: write-through (sc1) 16-byte
// slab stores, every storing wave's vmcnt(0), workgroup barrier, ONE relaxed agent-scope flag store; consumer: one lane
// polls relaxed, ONE agent-scope acquire, vmcnt(0), workgroup barrier, plain loads.  Spans are aligned to tile boundaries at
// the 8 chunk starts, so a hand-off never crosses the blockIdx % 8 classes (one XCD under round-robin placement).
//
// Staging schedule = gemm8p_kernel's, over units instead of k tiles (unit j in buffer j & 1, one half-tile per phase):
//   P1(j): A1(j+1)      P2(j): A0(j+2)      P3(j): W0(j+2)      P4(j): W1(j+2); s_waitcnt vmcnt(6)
// so the wait in P4(j) retires all of unit j + 1, which is first read one barrier later.  (Issuing W1 and A1 of unit
// j + 2 together in P4 -- simpler bookkeeping -- made P4's load segment longer than the partner wave's MFMA cluster:
// main loop 15 % slower.)  The staging cursor therefore moves to unit j + 2 between P1 and P2.
//
// Epilogue: the ring occupies 128 of the 160 KiB of LDS the whole time, so the staged epilogue works on 32-row x
// 32-column pieces (4 KiB per wave) and needs no row table: row -> (batch, token) is one reciprocal multiply + fix-up.
#include "gemm_device.h"

namespace cwm {

namespace {
struct SkCursor {
    int tile, kt, kend;
};
}  // namespace

// (the epilogue kind is a template parameter: with a run-time switch the 8 unrolled pieces x 4 kinds made a control-flow
// graph across which hipcc spilled ~230 registers, accumulators included)
template <int PLANES, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_sk_kernel(const GemmParams p) {
    constexpr int HALF_BYTES = 128 * 128;      // 128 rows x 128 B
    constexpr int BUF_BYTES = 4 * HALF_BYTES;  // A0 | A1 | W0 | W1
    constexpr int RING_BYTES = 2 * BUF_BYTES;
    constexpr int PIECE_BYTES = 32 * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int nk = p.K / (64 / PLANES);
    const int tiles_m = (p.M + 255) / 256;
    const int tiles_n = (p.N + 255) / 256;
    const int ntiles = tiles_m * tiles_n;

    // ---- this workgroup's units: a span of the stream-K region (tiles [0, S)), then whole tiles of the data-parallel rounds ----
    const int G = gridDim.x, per = G >> 3;  // workgroups per chunk (grid is a multiple of 8)
    const int chunk = blockIdx.x & 7, wi = blockIdx.x >> 3;
    const int rem = ntiles % G;
    const int S = (rem == 0) ? G : G + rem;   // G <= S < 2 G tiles are cut into equal spans
    const int dp_rounds = (ntiles - S) / G;   // then every workgroup takes one whole tile per round
    const int tc0 = (int)((int64_t)chunk * S / 8), tc1 = (int)((int64_t)(chunk + 1) * S / 8);
    const int cu = (tc1 - tc0) * nk;
    const int u0 = tc0 * nk + (int)((int64_t)wi * cu / per), u1 = tc0 * nk + (int)((int64_t)(wi + 1) * cu / per);
    const int n_units = u1 - u0 + dp_rounds * nk;
    const int ta = u0 / nk, ka = u0 - ta * nk;                       // first tile; its k range starts at ka (> 0: tail part)
    const int tb = (u1 - 1) / nk, kb_end = (u1 - 1) - tb * nk + 1;   // last tile; its k range ends at kb_end (< nk: head part)
    const int slab_id = chunk * per + wi;
    auto advance = [&](SkCursor& c) {
        if (++c.kt == c.kend) {
            if (c.tile < S && c.tile > ta) {  // next (lower) tile of the span
                --c.tile;
                c.kt = (c.tile == ta) ? ka : 0;
            } else {                          // span done: data-parallel rounds, tile ids S + round * G + slab_id
                c.tile = (c.tile < S) ? S + slab_id : c.tile + G;
                c.kt = 0;
            }
            c.kend = nk;
        }
    };
    auto tile_origin = [&](int id, int& m0, int& n0) {
        constexpr int GROUP_M = 4;
        const int group_sz = GROUP_M * tiles_n;
        const int g = id / group_sz;
        const int first_m = g * GROUP_M;
        const int gm = min(tiles_m - first_m, GROUP_M);
        const int in_g = id - g * group_sz;
        m0 = (first_m + (in_g % gm)) * 256;
        n0 = (in_g / gm) * 256;
    };

    // ---- LDS-DMA sources of the tile being staged: wave w stages pieces 2w, 2w+1 (rows 16w .. 16w+15) of every half-tile ----
    unsigned src[4][2];
    auto set_stage_tile = [&](int id) {
        int m0, n0;
        tile_origin(id, m0, n0);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int row = wave * 16 + jj * 8 + lane / 8;
                const int logical = (lane % 8) ^ lds_swizzle<64>(row);
                src[h][jj] = (unsigned)min(m0 + h * 128 + row, p.M - 1) * (unsigned)(p.lda * PLANES) + logical * 8;
                src[2 + h][jj] = (unsigned)(n0 + h * 128 + row) * (unsigned)(p.K * PLANES) + logical * 8;
            }
    };
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

    // ---- fragment addresses ----
    const int frow = lane & 15, fq = lane >> 4;
    int a_base[2], b_base[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        a_base[h] = lds_off<64>(wr * 64 + frow, h * 4 + fq);
        b_base[h] = 2 * HALF_BYTES + lds_off<64>(wc * 32 + frow, h * 4 + fq);
    }
    bf16x8 af[2][4], bw0[2][2], bw1[2][2];
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
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
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
#define CWM_PHASE_BARRIER()                \
    do {                                   \
        __builtin_amdgcn_sched_barrier(0); \
        __builtin_amdgcn_s_barrier();      \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)

    // ---- row -> (batch, token) without an integer division: m < 2^24, so one reciprocal multiply is off by at most one ----
    auto row_info = [&](int m) -> int2 {
        if (m >= p.M) return make_int2(-1, 0);
        int b = 0, tok = m, out_row = m, res_row = m;
        if (p.rows_in > 0) {
            b = (int)((float)m * p.rows_in_inv);
            tok = m - b * p.rows_in;
            if (tok < 0) {
                --b;
                tok += p.rows_in;
            } else if (tok >= p.rows_in) {
                ++b;
                tok -= p.rows_in;
            }
            out_row = b * p.rows_out + tok + p.out_row_offset;
            res_row = out_row;
            if constexpr (EPI == EPI_F32) {
                if (p.resid_rowmap) res_row = p.resid_rowmap[b * p.map_stride + tok];
            }
        }
        if constexpr (EPI == EPI_QKV) return make_int2(b * p.heads * p.n_tok + tok, 0);
        else return make_int2(out_row, res_row);
    };

    // ---- epilogue of one finished tile: eight 32-row x 32-column pieces through the wave's 4-KiB LDS buffer ----
    // Same access discipline as epilogue_piece_seq (gemm_device.h): unconditional GLOBAL loads / stores (lanes with nothing to
    // write go to the trash buffer), every bias load first, the next piece's residual rows requested before this piece's
    // stores are issued -- so the only vector-memory waits are counted ones for loads, never a drain of the stores.
    auto epilogue_tile = [&](int m0, int n0) {
        // (opaque copy of the lane id: keeps hipcc from hoisting the epilogue's address arithmetic out of the persistent
        // loop, where it would be spilled around the 224-register main loop)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int frow = lane_e & 15, fq = lane_e >> 4;
        const int lane = lane_e;
        char* wl = smem + RING_BYTES + wave * PIECE_BYTES;
        constexpr bool f32_out = EPI == EPI_F32;
        constexpr bool wide = f32_out || PLANES == 2;  // 8 chunks (128 B) per row; else 4 chunks (64 B)
        constexpr int NS = wide ? 4 : 2;               // read-back iterations per piece
        typedef __attribute__((address_space(1))) f32x4 gf32x4;
        gf32x4* const trash = (gf32x4*)(g_epilogue_trash + lane * 4);
        // per column block qn: QKV destination (which third, head, first d), uniform
        int which[2] = {0, 0}, qh[2] = {0, 0}, qd[2] = {0, 0};
        bf16* qbase[2] = {nullptr, nullptr};
        if constexpr (EPI == EPI_QKV) {
#pragma unroll
            for (int qn = 0; qn < 2; ++qn) {
                const int nb = n0 + qn * 128 + wc * 32;
                which[qn] = nb / p.qkv_dim;
                const int cD = nb - which[qn] * p.qkv_dim;
                qh[qn] = cD / p.head_dim;
                qd[qn] = cD - qh[qn] * p.head_dim;
                qbase[qn] = qkv_out_base(p, which[qn]);
            }
        }
        f32x4 bias4[2][2];
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + qn * 128 + wc * 32 + j * 16;
                bias4[qn][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.bias && n < p.N) bias4[qn][j] = *reinterpret_cast<const f32x4*>(p.bias + n + fq * 4);
            }
        // piece u = 2 blk + qn, row block blk = 2 qm + ih (32 rows of the wave's 128)
        auto block_rows = [&](int blk, int2 (&info)[NS]) {
            const int rowbase = m0 + (blk >> 1) * 128 + wr * 64 + (blk & 1) * 32;
#pragma unroll
            for (int s = 0; s < NS; ++s) info[s] = row_info(rowbase + (wide ? s * 8 + (lane >> 3) : s * 16 + (lane >> 2)));
        };
        auto load_resid = [&](int qn, const int2 (&info)[NS], f32x4 (&rv)[NS]) {
            const int n = n0 + qn * 128 + wc * 32 + (lane & 7) * 4;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const bool ok = p.resid && info[s].x >= 0 && n < p.N;
                rv[s] = *(ok ? (const gf32x4*)(p.resid + (size_t)info[s].y * p.ldr + n) : trash);  // raw; selected at use
            }
        };
        int2 info[NS];
        f32x4 rv[NS];
        block_rows(0, info);
        if constexpr (f32_out) load_resid(0, info, rv);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            __builtin_amdgcn_sched_barrier(0);  // one piece at a time: keeps the epilogue's register footprint beside the accumulators small
            const int blk = u >> 1, qn = u & 1, qm = blk >> 1, ih = blk & 1;
            const int nb = n0 + qn * 128 + wc * 32;
            // accumulators (+ bias, activation, split) -> piece buffer
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int il = 0; il < 2; ++il) {
                    const int r = il * 16 + frow;
                    f32x4 v = acc[qm][qn][ih * 2 + il][j] + bias4[qn][j];
                    if constexpr (f32_out) {
                        *reinterpret_cast<f32x4*>(wl + r * 128 + (((j * 4 + fq) ^ (r & 7)) << 4)) = v;
                    } else {
                        if constexpr (EPI == EPI_BF16_GELU) {
#pragma unroll
                            for (int e = 0; e < 4; e += 4) v = gelu_erf4(v);
                        } else if constexpr (EPI == EPI_QKV) {
                            if (which[qn] == 0) v *= p.q_scale;
                        }
                        bf16x4 hv, lv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const bf16 hi = (bf16)v[e];
                            hv[e] = hi;
                            lv[e] = (bf16)(v[e] - (float)hi);
                        }
                        const int ch = j * 2 + (fq >> 1), sub = (fq & 1) * 8;
                        *reinterpret_cast<bf16x4*>(wl + r * 128 + ((ch ^ (r & 7)) << 4) + sub) = hv;
                        if constexpr (PLANES == 2) *reinterpret_cast<bf16x4*>(wl + r * 128 + (((ch + 4) ^ (r & 7)) << 4) + sub) = lv;
                    }
                }
            }
            if (p.debug & 1) continue;
            // piece buffer -> registers (row-major 16-byte chunks), residual added; destinations
            f32x4 v[NS];
            gf32x4* dst[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if constexpr (wide) {
                    const int c = lane & 7, r = s * 8 + (lane >> 3);
                    v[s] = *reinterpret_cast<const f32x4*>(wl + r * 128 + ((c ^ (r & 7)) << 4));
                    if constexpr (f32_out) {
                        const int n = nb + c * 4;
                        const bool ok = info[s].x >= 0 && n < p.N;
                        if (ok && p.resid) v[s] += rv[s];
                        dst[s] = ok ? (gf32x4*)(p.C + (size_t)info[s].x * p.ldc + n) : trash;
                    } else {
                        const int n = nb + (c & 3) * 8, lo = c >> 2;
                        bf16* d;
                        if constexpr (EPI == EPI_QKV)
                            d = qbase[qn] + (size_t)lo * p.qk_plane + ((size_t)(info[s].x + qh[qn] * p.n_tok)) * p.head_dim + qd[qn] + (c & 3) * 8;
                        else
                            d = p.out_hi + a_pos<2>(info[s].x, p.ldo, n) + lo * kLoOffset;
                        dst[s] = (info[s].x >= 0 && n < p.N) ? (gf32x4*)d : trash;
                    }
                } else {
                    const int c = lane & 3, r = s * 16 + (lane >> 2);
                    v[s] = *reinterpret_cast<const f32x4*>(wl + r * 128 + ((c ^ (r & 7)) << 4));
                    const int n = nb + c * 8;
                    bf16* d;
                    if constexpr (EPI == EPI_QKV)
                        d = qbase[qn] + ((size_t)(info[s].x + qh[qn] * p.n_tok)) * p.head_dim + qd[qn] + c * 8;
                    else
                        d = p.out_hi + (size_t)info[s].x * p.ldo + n;
                    dst[s] = (info[s].x >= 0 && n < p.N) ? (gf32x4*)d : trash;
                }
            }
            // next piece: its row block's infos and residual rows are requested before this piece's stores go out
            if (u + 1 < 8) {
                if (((u + 1) >> 1) != blk) block_rows((u + 1) >> 1, info);
                if constexpr (f32_out) load_resid((u + 1) & 1, info, rv);
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) *dst[s] = v[s];
        }
    };

    // ---- split tiles: fp32 slab of this workgroup's accumulators, [wave][fragment][lane] x 16 bytes ----
    auto slab_ptr = [&](int id) {
        int lane_s = lane;  // opaque, as in epilogue_tile
        asm volatile("" : "+v"(lane_s));
        return p.sk_slabs + (size_t)id * (256 * 256) + (size_t)(wave * 32) * 256 + lane_s * 4;
    };
    auto publish_head = [&]() {
        // write-through (sc1) 16-byte slab stores: no release fence (an agent-scope release writes back the XCD's whole L2,
        // which at this point holds the other workgroups' freshly written output tiles)
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.sk_slabs, 0, (int)((size_t)gridDim.x * 256 * 256 * 4), 0x00020000);
        int lane_s = lane;  // opaque, as in epilogue_tile
        asm volatile("" : "+v"(lane_s));
        const unsigned off0 = (unsigned)slab_id * (256u * 256u * 4u) + (unsigned)(wave * 32) * 1024u + (unsigned)lane_s * 16u;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[a][b][i][j]), rsrc,
                                                               off0 + (unsigned)((((a * 2 + b) * 4 + i) * 2 + j) * 1024), 0, 16 /* sc1 */);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CWM_PHASE_BARRIER();
        // waves 4-7 run one barrier behind waves 0-3: when wave 4 has passed ITS barrier, all eight waves have drained their stores
        if (tid == 256) __hip_atomic_store(p.sk_flags + slab_id, p.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto reduce_tail = [&]() {
        // waves 0-3 lead: wave 0 acquires before ITS barrier, which every other wave's loads follow
        if (tid == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(p.sk_flags + slab_id - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.sk_epoch) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 24)) {  // several seconds: report instead of hanging the GPU
                    atomicExch(p.sk_err, 1u);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        CWM_PHASE_BARRIER();
        const float* sl = slab_ptr(slab_id - 1);
        // one quadrant (8 fragments = 32 registers in flight) at a time: 32 loads at once would not fit beside the accumulators
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                f32x4 t[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) t[i][j] = *reinterpret_cast<const f32x4*>(sl + (((a * 2 + b) * 4 + i) * 2 + j) * 256);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][b][i][j] += t[i][j];
                __builtin_amdgcn_sched_barrier(0);
            }
    };

    // ---- prologue: all of unit 0, then A0 / W0 / W1 of unit 1 ----
    SkCursor sc{tb, (tb == ta) ? ka : 0, kb_end};  // staging cursor
    SkCursor cc = sc;                              // compute cursor
    set_stage_tile(sc.tile);
    stage(H_A0{}, 0, sc.kt);
    stage(H_W0{}, 0, sc.kt);
    stage(H_W1{}, 0, sc.kt);
    stage(H_A1{}, 0, sc.kt);
    if (n_units > 1) {
        const int prev = sc.tile;
        advance(sc);
        if (sc.tile != prev) set_stage_tile(sc.tile);
        stage(H_A0{}, 1, sc.kt);
        stage(H_W0{}, 1, sc.kt);
        stage(H_W1{}, 1, sc.kt);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    CWM_PHASE_BARRIER();
    if (wr == 1) CWM_PHASE_BARRIER();  // the second wave group runs one barrier behind the first

    int m0c, n0c;
    tile_origin(cc.tile, m0c, n0c);
    zero_acc();
    for (int j = 0; j < n_units; ++j) {
        const int buf = j & 1;
        const char* base = smem + buf * BUF_BYTES;
        const bool has2 = j + 2 < n_units;
        // ---- P1: quadrant (0, 0); last half-tile of unit j + 1 (the staging cursor still points at it) ----
        read_w(base, 0, bw0);
        read_a(base, 0);
        if (j + 1 < n_units) stage(H_A1{}, buf ^ 1, sc.kt);
        CWM_PHASE_BARRIER();
        mfma_quadrant(acc[0][0], bw0);
        CWM_PHASE_BARRIER();
        if (has2) {  // staging cursor -> unit j + 2
            const int prev = sc.tile;
            advance(sc);
            if (sc.tile != prev) set_stage_tile(sc.tile);
        }
        // ---- P2: quadrant (0, 1) ----
        read_w(base, 1, bw1);
        if (has2) stage(H_A0{}, buf, sc.kt);
        CWM_PHASE_BARRIER();
        mfma_quadrant(acc[0][1], bw1);
        CWM_PHASE_BARRIER();
        // ---- P3: quadrant (1, 1) ----
        read_a(base, 1);
        if (has2) stage(H_W0{}, buf, sc.kt);
        CWM_PHASE_BARRIER();
        mfma_quadrant(acc[1][1], bw1);
        CWM_PHASE_BARRIER();
        // ---- P4: quadrant (1, 0); retire unit j + 1 ----
        if (has2) {
            stage(H_W1{}, buf, sc.kt);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        CWM_PHASE_BARRIER();
        mfma_quadrant(acc[1][0], bw0);
        CWM_PHASE_BARRIER();

        // ---- end of this tile's k range in the span ----
        if (cc.kt + 1 == cc.kend) {
            if (cc.kend < nk) {
                publish_head();  // the tail of this tile belongs to the next workgroup, which finishes it
            } else if (!(p.debug & 2)) {
                // (two copies of the epilogue rather than `if (tail) reduce; epilogue`: merging 128 conditionally updated
                // accumulator registers at the join made hipcc copy and spill half of them)
                if (cc.tile == ta && ka > 0) {
                    reduce_tail();  // the head came from the previous workgroup
                    epilogue_tile(m0c, n0c);
                } else {
                    epilogue_tile(m0c, n0c);
                }
            } else {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int jx = 0; jx < 2; ++jx) asm volatile("" ::"v"(acc[a][b][i][jx]));
            }
            zero_acc();
            const int prev = cc.tile;
            advance(cc);
            if (cc.tile != prev && j + 1 < n_units) tile_origin(cc.tile, m0c, n0c);
        } else {
            ++cc.kt;
        }
    }
    if (wr == 0) CWM_PHASE_BARRIER();  // balance the stagger barrier
#undef CWM_PHASE_BARRIER
}

// Host side: one workspace (slabs + flags) per process, grown on demand.  Launches that use it are ordered by the
// stream they are enqueued on; the library enqueues a model's kernels on one stream at a time.
namespace {
struct SkWorkspace {
    float* slabs = nullptr;
    unsigned* flags = nullptr;  // [grid] flags, then the error word
    int grid = 0;
    unsigned epoch = 0;
};
SkWorkspace g_sk;
}  // namespace

int sk_grid_size() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus = prop.multiProcessorCount;
    }
    return cus & ~7;
}

// Can (M, N, K) run on the stream-K kernel with `grid` workgroups?  Every span must hold at least one whole tile.
bool sk_shape_ok(int M, int N, int K, int planes, int grid) {
    if (grid < 8 || M >= (1 << 24)) return false;
    const int nk = K / (64 / planes);
    const int64_t ntiles = (int64_t)((M + 255) / 256) * ((N + 255) / 256);
    if (nk < 1 || ntiles < grid || ntiles * nk >= (1ll << 30)) return false;
    const int per = grid / 8;
    const int64_t S = (ntiles % grid == 0) ? grid : grid + ntiles % grid;
    for (int c = 0; c < 8; ++c) {
        const int64_t t0 = c * S / 8, t1 = (c + 1) * S / 8;
        if ((t1 - t0) < per) return false;  // every span holds at least one tile: (t1 - t0) * nk / per >= nk
    }
    return true;
}

int launch_gemm_sk(GemmParams& p, int planes, hipStream_t stream) {
    const int grid = sk_grid_size();
    CWM_REQUIRE(sk_shape_ok(p.M, p.N, p.K, planes, grid), "gemm_sk: shape M=%d N=%d K=%d does not fill %d workgroups", p.M, p.N, p.K, grid);
    if (g_sk.grid < grid) {
        if (g_sk.slabs) (void)hipFree(g_sk.slabs);
        if (g_sk.flags) (void)hipFree(g_sk.flags);
        g_sk = SkWorkspace();
        CWM_HIP_CHECK(hipMalloc((void**)&g_sk.slabs, (size_t)grid * 256 * 256 * sizeof(float)));
        CWM_HIP_CHECK(hipMalloc((void**)&g_sk.flags, (size_t)(grid + 1) * sizeof(unsigned)));
        CWM_HIP_CHECK(hipMemset(g_sk.flags, 0, (size_t)(grid + 1) * sizeof(unsigned)));
        g_sk.grid = grid;
    }
    if (++g_sk.epoch == 0) ++g_sk.epoch;  // 0 is the "never published" value of a fresh flag
    p.sk_slabs = g_sk.slabs;
    p.sk_flags = g_sk.flags;
    p.sk_err = g_sk.flags + g_sk.grid;
    p.sk_epoch = g_sk.epoch;
    p.rows_in_inv = p.rows_in > 0 ? 1.0f / (float)p.rows_in : 0.f;
    typedef void (*kern_t)(const GemmParams);
    static const kern_t ks[2][4] = {
        {gemm_sk_kernel<1, EPI_F32>, gemm_sk_kernel<1, EPI_BF16_GELU>, gemm_sk_kernel<1, EPI_BF16>, gemm_sk_kernel<1, EPI_QKV>},
        {gemm_sk_kernel<2, EPI_F32>, gemm_sk_kernel<2, EPI_BF16_GELU>, gemm_sk_kernel<2, EPI_BF16>, gemm_sk_kernel<2, EPI_QKV>}};
    CWM_REQUIRE(p.epi >= 0 && p.epi < 4, "gemm_sk: bad epilogue kind %d", p.epi);
    const size_t smem = 2 * 4 * 128 * 128 + 8 * 32 * 128;  // ring + piece buffers = 160 KiB
    kern_t k = ks[planes - 1][p.epi];
    if (int rc = cwm_set_max_lds((const void*)k, (int)smem)) return rc;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), smem, stream, p);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

// 1 if a split-tile hand-off of any stream-K launch so far timed out (development / tests)
int sk_error_flag() {
    if (!g_sk.flags) return 0;
    unsigned v = 0;
    if (hipMemcpy(&v, g_sk.flags + g_sk.grid, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)v;
}

}  // namespace cwm
