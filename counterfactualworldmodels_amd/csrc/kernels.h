// Internal launch interfaces of the HIP kernels (not part of the C ABI; see include/cwm_hip.h).
#pragma once
#include "common.h"

namespace cwm {

// Execution options of one model handle (cwm_model_set_option / cwm_conj_set_option) -- every switch that was a process-wide global of the library
// until round 4.  The defaults are the measured best; the other values exist for same-box A/B measurements and for the bitwise cross-checks of the
// test suite.  A launch reads them through its parameter struct (GemmParams.tune, AttnParams.tune; nullptr = the defaults), a forward through its
// Engine -- never from a global, so two models in one process cannot change each other's kernels.
struct Tuning {
    int gemm_tile = 0;     // 0 automatic per shape (gemm_choose_tile), 1: 128x128, 4: 256x256 8-phase, 6: 8-phase rounds + 128x128 remainder rows
    int gemm_debug = 0;    // bit mask of ablations / A-B switches: 1 skip the epilogue's global stores, 2 skip the epilogue, 4 no 4-stage ring for small launches,
                           // 8 skip every LayerNorm launch (timing only), 32 no split-K, 128 the one-lane tile choice also inside a two-lane call, 256 small launches keep
                           // 128-row tiles where the default takes 64x128 ones, 512 bf16-output GEMMs with K < 512 stay on 128x128 tiles, 1024 no half-width column tiles in the
                           // 8-phase kernel (and N = 384 back on 128x128 tiles)
    int gemm_staged = 1;   // 0: the per-fragment epilogue of round 1 everywhere (it stays the fallback for unaligned widths)
    int gemm_direct = 1;   // 1: bf16-output epilogues store 16 bytes per lane straight from the accumulators; 2: the fp32-output ones too; 0: LDS-staged everywhere
    int attn_kernel = 0;   // 0 automatic, 1: 4-wave kernel (attention.hip), 3: software-pipelined kernel (attention_pipe.hip)
    int attn_remap = 1;    // 0: plain workgroup order instead of one XCD per (batch, head) with the ragged query tiles last
    int attn_tail = 1;     // 0: the regular schedule also for a ragged last query tile of <= 32 rows (attention_tail.h)
    int attn_ksplit = 1;   // 0: a nearly empty last round of workgroups runs its items whole instead of cutting them into key ranges
    int index_fused = 1;       // 0: the index prologue as the four launches of rounds 1-4 (memset, mask_to_perm, patch_gather, perm_to_rank) instead of one
    int prune_last_block = 1;  // 0: the last decoder block runs over all tokens
    int min_lane_rows = 0;     // encoder rows per half batch from which a forward splits into two lanes; 0: the model family's default (engine.h)
    int conj_ctx_stream = 1;   // 0: the IMU-conditioned model's context stream on the lane's own stream
    int conj_attn = 1;         // 0: the fp32 VALU cross / context attention kernels instead of the MFMA ones
    // development library only (csrc/dev.hip, cwm_gemm_tile_override): per-shape tile configuration, 0 = no opinion
    int (*tile_hook)(int M, int N, int K, int epi, int overlapped) = nullptr;
};
const Tuning& default_tuning();
// What the stand-alone entry points (cwm_linear, cwm_attention ...) use and what a new model handle starts from: the defaults, unless the
// development library's cwm_debug_set changed this THREAD's copy (libcwm_hip.so exports no way to)
Tuning& thread_tuning();
int tuning_set(Tuning& t, const char* key, int value);  // 0, or -1 for an unknown key
int tuning_set_production(Tuning& t, const char* key, int value);  // the production setters: -2 for the timing-only ablation bits of "gemm_debug" (engine.hip)
int tuning_get(const Tuning& t, const char* key, int* value);

enum GemmEpilogue : int {
    EPI_F32 = 0,        // C = acc + bias (+ resid[rowmap])            fp32 out
    EPI_BF16_GELU = 1,  // out = split_bf16(gelu_erf(acc + bias))      bf16 plane(s) out
    EPI_BF16 = 2,       // out = split_bf16(acc + bias)                bf16 plane(s) out
    EPI_QKV = 3,        // per-head scatter: Q (scaled), K, V -> [B*H,N,hd]
};

struct GemmParams {
    // operands (bf16, K-contiguous) in the A-operand layout of the mode (common.h a_pos): fast A[M][lda], W[Npad][K];
    // parity A[M][2*lda], W[Npad][2*K] with hi/lo interleaved per 32-k block.  lda / K are LOGICAL k counts.
    const bf16* A;
    const bf16* W;
    int lda;
    int M, N, K;
    int m_offset;       // this launch covers rows [m_offset, m_offset + M) of the problem (launch_gemm's mixed-tile split); usually 0
    const float* bias;  // [N] or nullptr
    int epi;
    // row mapping (rows_in == 0: identity).  m = b*rows_in + i  ->  out row b*rows_out + i,
    // residual row = resid_rowmap ? resid_rowmap[b*map_stride + i] : out row
    int rows_in, rows_out, map_stride;
    int out_row_offset;  // added to the out row (and to the default residual row): writes a per-sample row block (last decoder block)
    const int* resid_rowmap;
    // EPI_F32
    float* C;
    int ldc;
    const float* resid;
    int ldr;
    // EPI_BF16*
    bf16* out_hi;
    int64_t out_plane;
    int ldo;
    // EPI_QKV
    bf16* q_out;
    bf16* k_out;
    bf16* v_out;
    int64_t qk_plane;
    int qkv_dim, heads, head_dim, n_tok;
    float q_scale;
    // split-K of the latency-bound small launches (gemm.hip, deep-ring 128x128 kernel); filled in by launch_gemm
    int splitk;              // K is cut into this many ranges, one workgroup each (1: off)
    float* sk2_slabs;        // [tiles * splitk][128 * 128] fp32 partial accumulators
    unsigned* sk2_count;     // [tiles] arrival counters (0 between launches)
    int staged;  // set by launch_gemm: epilogue through LDS with full-line global accesses (gemm.hip)
    int direct;  // set by launch_gemm: 16-byte stores straight from the accumulators, W tile staged with permuted rows (gemm_device.h epilogue_direct)
    int overlapped;  // set by the engine: the launch runs beside another lane's kernels, so a partly filled last round of workgroups is not lost
    int debug;  // set by launch_gemm from tune->gemm_debug (the kernels read bits 0 / 1: skip the epilogue's global stores / the epilogue)
    const Tuning* tune;  // execution options of the calling model (nullptr: defaults)
};

constexpr int kSplitKSlots = 512;  // >= CUs: a split launch has at most one part per CU
int splitk_workspace_alloc(float** slabs, unsigned** counts, hipStream_t stream);  // counters zeroed on `stream`
int gemm_splitk_parts(const GemmParams& p, int planes);  // K ranges the deep-ring 128x128 kernel would cut this launch into (1: no split-K)
int launch_gemm(const GemmParams& p, int planes, hipStream_t stream);
int gemm_choose_tile(const GemmParams& p, int planes);
bool gemm_mixed_split(const GemmParams& p, GemmParams* big, GemmParams* rest);  // tile configuration 6
int launch_gemm_tile(const GemmParams& p, int planes, int cfg, hipStream_t stream);  // 1: 128x128, 4: 256x256 8-phase, 6: mixed (Tuning.gemm_tile)
int gemm_cu_count();  // compute units of the current device, rounded down to a multiple of the 8 XCDs (256 on MI355X)
int gemm_prof_dump();  // builds with -DCWM_GEMM_PROF: per-workgroup timers of gemm8p_kernel -> /tmp/gemm_blocks.bin

struct AttnParams {
    const bf16* q;   // [planes][B*H][N][64]   (q pre-scaled by hd^-0.5)
    const bf16* k;   // [planes][B*H][N][64]
    const bf16* v;   // [planes][B*H][N][64]  (row-major like K; transposed on the LDS read)
    int64_t qk_plane;
    bf16* o;         // [planes][B*N][ldo]  (head h at columns h*64..)
    int64_t o_plane;
    int ldo;
    int n_tok, heads, batch;
    int remap;       // set by launch_attention: XCD-aware workgroup -> (query tile, head) mapping (attention_device.h; "attn_remap" switch)
    int q_off, n_q;  // queries = rows [q_off, q_off + n_q) of every (batch, head); n_q == 0: all n_tok.  O rows are b * n_q + (q - q_off)
    // key-split tail round (attention_pipe.hip, "attn_ksplit" switch), set by launch_attention_pipe: work items [0, ks_main) run whole; each of the
    // remaining items is cut into ks_parts key ranges, one workgroup each, which leave (O^T unnormalised, max, sum) in ks_scratch for
    // attention_combine_kernel.  ks_parts == 0: off
    int ks_main, ks_parts, ks_nqb;
    float* ks_scratch;  // [items - ks_main][ks_parts][128 queries][68]: O[64], max, sum, -, -
    int tail_split;  // set by launch_attention: a ragged last query tile of at most 32 rows splits the KEYS over its four waves (attention_tail.h; "attn_tail" switch)
    const Tuning* tune;  // execution options of the calling model (nullptr: defaults)
};

int launch_attention(const AttnParams& p, int planes, hipStream_t stream);
int attention_pipe_prof(int i);  // per-phase s_memtime totals of block 0 wave 0 (builds with -DCWM_ATTN_PROF only)
int launch_attention_pipe(const AttnParams& p, int planes, hipStream_t stream);  // attention_pipe.hip; arguments checked by launch_attention

struct LayerNormParams {
    const float* x;  // rows of length D, row stride ldx
    int ldx;
    const float* gamma;
    const float* beta;
    float eps;
    int D;
    int rows;        // number of output rows
    // input row for output row r: (r / rows_out_per_b) * rows_in_per_b + in_offset + (r % rows_out_per_b)
    // (rows_out_per_b == 0: identity)
    int rows_out_per_b, rows_in_per_b, in_offset;
    bf16* out;       // [planes][rows][ldo]
    int64_t out_plane;
    int ldo;
    float* out_f32;  // optional fp32 copy of the normalised rows ([rows][D]); may be nullptr
};

int launch_layernorm(const LayerNormParams& p, int planes, hipStream_t stream);

// mask[B,Nt] (1 = masked) -> perm[B,Nt] = [visible tokens ascending | masked tokens ascending];
// err[0] is set to 1 if any row's visible count != n_vis.
int launch_mask_to_perm(const uint8_t* mask, int B, int Nt, int n_vis, int* perm, int* err, hipStream_t stream);
// RectangularizeMasks on device masks (elementwise.hip): masked count per row; apply the host's picks [R | rows | offsets | to_value | picks] in place
int launch_mask_row_counts(const uint8_t* mask, int B, int Nt, int* counts, hipStream_t stream);
int launch_mask_flip_picks(uint8_t* mask, int Nt, const int* table, int n_rows, hipStream_t stream);

struct PatchGatherParams {
    const float* x;  // frames; element (b,c,t,y,x) at b*sb + c*sc + t*st + y*W + x
    int64_t sb, sc, st;
    int normalize;   // apply (x - mean_c)/std_c in-kernel (prediction.py:309-310)
    int C, H, W, P;
    const int* perm; // [B][perm_stride]
    int Nt, n_rows;  // real tokens per sample; rows per sample to gather (= n_vis)
    int perm_stride; // 0 = Nt; padded predictors: Nt + max_padding_tokens (entries >= Nt are pad slots)
    int B;
    bf16* out;       // [planes][B*n_rows][ld]  patch vector order (c, ph, pw), zero-padded to ld
    int64_t out_plane;
    int ld;
};

int launch_patch_gather(const PatchGatherParams& p, int planes, hipStream_t stream);
// mask[B][L] (L = perm_stride or Nt) -> perm[B][L], rank[B][L] (inverse; may be nullptr), err_rows[b] = (visible count of row b != n_vis), and the
// gather of the first n_rows visible tokens of every sample, in one launch (elementwise.hip index_gather_kernel); p.perm is not read
int launch_index_gather(const PatchGatherParams& p, const uint8_t* mask, int n_vis, int* perm, int* rank, int* err_rows, int planes, hipStream_t stream);

// x_full[b][n_vis + j][:] = mask_token + pos[perm[b][n_vis + j]]   (vmae.py:556-557)
int launch_fill_mask_tokens(float* x_full, const float* mask_token, const float* pos, const int* perm, int B, int Nt, int n_vis, int D, hipStream_t stream);

struct UnembedParams {
    const float* y;  // [B][Nm][P*P*C], feature order (ph, pw, c)
    const float* x;  // raw frames, element (b,t,c,y,x) at b*sb + t*st + c*sc + y*W + x
    int64_t sb, sc, st;
    const uint8_t* mask;  // [B][Nt]
    const int* rank;      // [B][Nt]: position of token tau in perm (>= n_vis for masked tokens)
    int B, T, C, H, W, P, n_vis, Nm;
    float* out;           // [B][T][C][H][W] contiguous
};

int launch_unembed(const UnembedParams& p, hipStream_t stream);

struct ShiftPromptParams {
    const float* x;         // [B][T][C][H][W] contiguous frames
    int B, S, T, C, H, W, P, frame, fix_passive;
    const uint8_t* active;  // [B*S][Nt], 0 at the active (moved) patches
    const uint8_t* masks;   // [B*S][Nt], 0 at the passive (kept visible) patches
    const int* shifts;      // [B*S][2] (dy, dx) in patch units
    float* x_out;           // [B*S][T][C][H][W]
    uint8_t* mask_out;      // [B*S][Nt]
};

int launch_shift_prompts(const ShiftPromptParams& p, hipStream_t stream);
int launch_prompt_table_expand(const int* table, int S, int n, int gw, int T, int frame, uint8_t* active, uint8_t* passive, int* shifts, hipStream_t stream);

// ---- IMU-conditioned conjoined predictor (conj_kernels.hip) ------------------------------------------
struct SmallAttnParams {
    const float* qkv;  // [B*n_tok][3*heads*head_dim] fp32 (bias already added; q NOT yet scaled)
    int B, n_tok, heads, head_dim;
    bf16* o;           // [planes][B*n_tok][ldo]
    int64_t o_plane;
    int ldo;
};
int launch_small_attention(const SmallAttnParams& p, int planes, hipStream_t stream);       // conj_kernels.hip: fp32 VALU form (any head_dim <= 64)
int launch_small_attention_mfma(const SmallAttnParams& p, int planes, hipStream_t stream);  // conj_attention.hip: head_dim 32
bool small_attention_mfma_ok(int n_tok, int head_dim);

// ext_mask[b] = [mask[b] | pad slot j masked unless j < vmax - visible(b)]  (conjoined_vmae.py:49-116)
int launch_pad_mask(const uint8_t* mask, int B, int N, int P, int vmax, uint8_t* ext_mask, hipStream_t stream);
// rows of x[B*n_rows][D] whose permutation entry is a pad slot (>= n_real) are set to `token` (null_token_enc)
int launch_fix_pad_rows(float* x, const int* perm, int B, int perm_stride, int n_rows, int n_real, int D, const float* token, hipStream_t stream);
// rows j of y[B][n_out][D] whose slot perm[b][n_vis + j] is a pad slot are zeroed (x * ~null_mask, conjoined_vmae.py:998-1002)
int launch_zero_pad_out_rows(float* y, const int* perm, int B, int perm_stride, int n_vis, int n_out, int n_real, int D, hipStream_t stream);

struct ImuGatherParams {
    const float* imu;  // [B][C][L]
    int B, C, L, tubelet;
    const int* perm;   // [B][perm_stride]
    int perm_stride, n_rows, n_real;
    bf16* out;         // [planes][B*n_rows][ld], K order (c, s), zero padded
    int64_t out_plane;
    int ld;
};
int launch_imu_gather(const ImuGatherParams& p, int planes, hipStream_t stream);

struct CrossAttnParams {
    const float* qk;      // [B*N][2D] main stream (fp32)                 -- VALU kernels (conj_kernels.hip)
    const float* v;       // [B*N][D]
    const bf16* qk_op;    // the same projections in the GEMM A-operand layout (common.h a_pos, row width 2D / D) -- MFMA kernel (conj_attention.hip)
    const bf16* v_op;
    const float* qk_src;  // [B*M][2D] context stream
    const float* v_src;   // [B*M][D]
    int B, N, M, heads, head_dim;  // D = heads*head_dim
    float scale;
    bf16* y;              // [planes][B*N][D]  main-stream update (softmax over the M context tokens)
    int64_t y_plane;
    bf16* y_src;          // [planes][B*M][D]  context update (softmax over the N main tokens)
    int64_t y_src_plane;
    float* scores_t;      // scratch [B][heads][M][N]
    float* partial;       // scratch, cross_attention_partial_floats(B, heads, M, head_dim) floats
};
int launch_cross_attention(const CrossAttnParams& p, int planes, hipStream_t stream);       // fp32 VALU kernels: qk / v fp32, scores_t + partial scratch
size_t cross_attention_partial_floats(int B, int heads, int M, int head_dim);
int launch_cross_attention_mfma(const CrossAttnParams& p, int planes, hipStream_t stream);  // MFMA kernel: qk_op / v_op, partial scratch
int launch_cross_attention_mfma_roles(const CrossAttnParams& p, int planes, hipStream_t stream_a, hipStream_t stream_b, int roles);  // bit 0: main update on stream_a, bit 1: context update on stream_b
bool cross_attention_mfma_ok(int head_dim, int M);
bool cross_attention_mfma_fits(int B, int N, int heads, int head_dim);  // the MFMA kernel's 32-bit offsets hold this lane (else: the VALU kernels)
size_t cross_attention_mfma_partial_floats(int B, int heads, int M, int head_dim);

int launch_perm_to_rank(const int* perm, int* rank, int B, int Nt, hipStream_t stream);

int launch_split_bf16(const float* x, int64_t n, bf16* hi, bf16* lo, hipStream_t stream);

}  // namespace cwm
