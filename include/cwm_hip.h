/*
 * cwm_hip.h -- C ABI of libcwm_hip.so: the MI355X (gfx950) VMAE predictor forward pass.
 *
 * This is the drop-in boundary for ONE path of neuroailab/CounterfactualWorldModels: everything
 * executed inside `self.predictor(x, mask)` at cwm/models/prediction.py:419-422, i.e.
 * `PretrainVisionTransformer.forward` (cwm/models/VideoMAE/vmae.py:539-560) with the primitives of
 * cwm/models/VideoMAE/utils.py (PatchEmbed :156-198, Attention :57-121, Mlp :37-54, Block :124-153),
 * plus the two thin wrapper steps either side of it: `_preprocess` / `imagenet_normalize`
 * (prediction.py:304-312, utils.py:15-21) and `pred_patches_to_video` (prediction.py:245-259).
 *
 * Conventions
 *   - plain C, no C++/torch types; every pointer named *_dev is a HIP device pointer owned by the
 *     caller; `stream` is a hipStream_t passed as void* (NULL = default stream)
 *   - functions return 0 on success, a negative code on failure, never throw; the message is
 *     available from cwm_last_error() (thread-local)
 *   - the library owns packed (bf16 hi/lo) weights and a workspace per model handle; no caller
 *     tensor is retained past a call
 *   - one process per GPU; a model handle lives on the device that was current at create time
 */
#ifndef CWM_HIP_H
#define CWM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The shared objects are built with -fvisibility=hidden: the entry points declared here (and, in libcwm_hip_dev.so, those of cwm_hip_dev.h)
 * are all they export. */
#define CWM_API __attribute__((visibility("default")))

#define CWM_OK 0
#define CWM_ERR_INVALID (-1) /* bad argument / precondition (mirrors the reference's shape errors) */
#define CWM_ERR_HIP (-2)     /* a HIP runtime call failed */
#define CWM_ERR_MASK (-3)    /* rows of `mask` do not all have n_vis visible tokens (vmae.py:167 reshape) */

/* Arithmetic mode of the GEMM / attention kernels. */
#define CWM_MODE_FAST 1   /* bf16 MFMA operands, fp32 accumulate                         */
#define CWM_MODE_PARITY 2 /* split-bf16 (hi+lo, 3 MFMAs per product): <=1e-3 vs fp32 CPU */

/* Mirrors the constructor arguments of PretrainVisionTransformer (vmae.py:261-297) that the
 * factories vmae.py:563-619 set; tubelet_size is 1 on this path. */
typedef struct cwm_config {
    int32_t img_h, img_w;   /* 224, 224 */
    int32_t patch;          /* 8 (base_8x8), 4 (large_4x4), 16 */
    int32_t num_frames;     /* 2 */
    int32_t in_chans;       /* 3 */
    int32_t enc_dim, enc_depth, enc_heads;
    int32_t dec_dim, dec_depth, dec_heads;
    int32_t mlp_ratio;      /* 4 */
    float ln_eps;           /* 1e-6 */
} cwm_config;

typedef struct cwm_model cwm_model;
struct cwm_kernel_stats;

/* replaces: model construction via the factories, vmae.py:597-619 */
CWM_API int cwm_model_create(const cwm_config* cfg, cwm_model** out);
CWM_API void cwm_model_destroy(cwm_model* m);

/* replaces: `model.load_state_dict(...)` (prediction.py:81-107).  `key` is the reference state-dict
 * name, `data` fp32 in PyTorch layout, on the host (on_device=0) or on the
 * model's device (on_device=1).  The tensor is copied and packed; the caller's buffer is not kept.
 * Re-loading a key overwrites it.  Unknown key or wrong shape -> CWM_ERR_INVALID. */
CWM_API int cwm_model_load_weight(cwm_model* m, const char* key, const float* data, int on_device, const int64_t* shape, int ndim);
/* Number of state-dict tensors not loaded yet (0 = ready); fills `buf` with the first missing key. */
CWM_API int cwm_model_missing_weights(cwm_model* m, char* buf, int buflen);

typedef struct cwm_forward_args {
    uint32_t struct_size;    /* sizeof(cwm_forward_args) of the header the CALLER was compiled against: fields added at the end by later versions
                              * are read only when the size covers them; a size below the first version's or above 4096 is CWM_ERR_INVALID (0.6 inserted this
                              * field FIRST: an ABI break against 0.5, whose callers must be rebuilt -- their x_dev pointer fails the upper bound) */
    /* frames: element (b, c, t, y, x) at x_dev[b*x_stride_b + c*x_stride_c + t*x_stride_t + y*W + x]
     * (so both [B,C,T,H,W] and the wrapper's [B,T,C,H,W] layouts are accepted without a copy) */
    const float* x_dev;
    int64_t x_stride_b, x_stride_c, x_stride_t;
    /* 1: x_dev is the raw [0,1] wrapper input and (x-mean)/std is fused into the frame load
     * (prediction.py:309-310); 0: x_dev is already what the model should see (vmae.py:539 seam) */
    int32_t normalize;
    const uint8_t* mask_dev; /* bool [B, Nt], 1 = masked (torch.bool storage) */
    int32_t batch;
    int32_t n_vis;           /* visible tokens per row; every row must have exactly this many */
    float* y_tokens_dev;     /* out [B, Nt - n_vis, in_chans*patch*patch] fp32, feature order (ph pw c); n_vis == Nt (nothing masked):
                              * [B, Nt, ...] = head(norm(x)) of every token, as vmae.py:250-253 (y_video_dev must then be NULL) */
    /* optional fused `pred_patches_to_video`: out [B, T, C, H, W]; visible patches are copied from
     * xraw_dev (same strides as x_dev; NULL = use x_dev, valid when normalize=1) */
    float* y_video_dev;
    const float* xraw_dev;
    int32_t mode;            /* CWM_MODE_FAST or CWM_MODE_PARITY */
    int32_t check;           /* 1: synchronise the stream and verify the mask precondition */
    void* stream;
} cwm_forward_args;

/* replaces: `self.predictor(self._preprocess(x), mask)` prediction.py:419-422
 *           = PretrainVisionTransformer.forward vmae.py:539-560,
 * and optionally `pred_patches_to_video` prediction.py:245-259. */
CWM_API int cwm_forward(cwm_model* m, const cwm_forward_args* args);

/* Batch lanes (no counterpart in the reference: an execution option of this library).  lanes = 2 (default): a call with
 * batch >= 2 whose halves keep >= 3000 encoder rows (ViT-B/8: batch >= 8; ViT-L/4: batch >= 2) runs as two half batches, the first on args->stream and the second on a stream owned by the model, forked and joined
 * with events inside cwm_forward, so the caller sees ordinary stream semantics; results are those of the single-lane call up to the
 * kernel choice per GEMM shape (fp32 re-association, < 1e-5).  lanes = 1: everything on args->stream. */
CWM_API int cwm_model_set_lanes(cwm_model* m, int lanes); /* 1 .. 4 (lane l owns batch rows [ceil(B l / n), ceil(B (l + 1) / n)); fewer lanes are used when a lane would keep < 3000 encoder rows); more than two lanes measured slower on MI355X */

/* Execution options of ONE model handle (no counterpart in the reference).  The defaults are the measured best and what production callers run;
 * the other values exist for same-box A/B measurements and for the bitwise cross-checks of the test suite.  Options are per handle: two models in
 * one process never see each other's settings (until round 4 these were process-wide switches).  Unknown key -> CWM_ERR_INVALID.  Results do not
 * depend on an option beyond fp32 re-association (< 1e-5).  (The timing-only ablation bits 1 / 2 / 8 of "gemm_debug" -- skip the epilogue's stores / the
 * epilogue / every LayerNorm: wrong outputs -- are REFUSED here with CWM_ERR_INVALID; the development library sets them through cwm_debug_set.)
 *   "gemm_tile"    0 automatic per shape, 1: 128x128 tiles, 4: 256x256 8-phase kernel, 6: 8-phase rounds + 128x128 remainder rows
 *   "gemm_direct"  1: bf16-output epilogues store 16 bytes per lane straight from the accumulators; 2: the fp32-output ones too; 0: LDS-staged
 *   "gemm_staged"  0: the per-fragment epilogue of round 1
 *   "gemm_debug"   bit mask of result-preserving A/B switches: 4 no 4-stage ring for small launches, 32 no split-K, 128 the one-lane tile choice also inside a two-lane call, 256 small launches
 *                  keep 128-row tiles, 512 bf16-output GEMMs with K < 512 stay on 128x128 tiles, 1024 no half-width column tiles in the 8-phase kernel
 *   "attn_kernel"  0 automatic, 1: 4-wave kernel, 3: software-pipelined kernel
 *   "attn_remap"   0: plain workgroup order instead of one XCD per (batch, head) with the ragged query tiles last
 *   "attn_tail"    0: the regular schedule also for a ragged last query tile of <= 32 rows
 *   "attn_ksplit"  0: a nearly empty last round of workgroups runs its items whole instead of cutting them into key ranges
 *   "index_fused"       0: the index prologue (mask -> permutation, its inverse, the row check, the patch gather) as the four launches of rounds 1-4
 *   "prune_last_block"  0: the last decoder block runs over all tokens
 *   "min_lane_rows"     encoder rows per half batch from which a forward splits into two lanes (0: the default, 3000 / 12000 for the IMU model)
 *   "conj_attn"         0: the fp32 VALU cross / context attention kernels of the IMU-conditioned model instead of the MFMA ones
 *   "conj_ctx_stream"   0: the IMU-conditioned model's context stream on the lane's own stream instead of a side stream */
CWM_API int cwm_model_set_option(cwm_model* m, const char* key, int value);

/* ---- IMU-conditioned conjoined padded predictor (BASELINE configs[4]) ------------------------------
 * replaces: ConjoinedPaddedVisionTransformer.forward for the `imu400_base_4x4patch_2frames_1tube` family
 * (cwm/models/VideoMAE/conjoined_vmae.py:889-1011, 852-887; factory :1230-1243), i.e. two token streams
 * (RGB + IMU) with null-token padding (:49-165), ImuEncoder tokenisation (:1013-1147) and
 * CrossAttentionTransformerBlock / BidirectionalCrossAttention (cwm/models/transformer.py:253-378, 442-583). */
typedef struct cwm_conj_config {
    cwm_config main;            /* RGB stream (patch 4, 768/12/12 - 384/6/4 for the shipped factory) */
    int32_t main_max_pad;       /* max_padding_tokens of the RGB stream (64) */
    int32_t ctx_in_chans, ctx_seq_len, ctx_tubelet;   /* 6, 400, 16 */
    int32_t ctx_enc_dim, ctx_dec_dim, ctx_enc_heads, ctx_dec_heads; /* 384, 192, 12, 6 */
    int32_t ctx_max_pad;        /* 25 */
    int32_t n_enc_cross, enc_cross[16]; /* cross block BEFORE these encoder layers (0,3,6,9) */
    int32_t n_dec_cross, dec_cross[16]; /* cross block AFTER these decoder layers (0,1,2,3) */
    int32_t cross_heads, cross_mlp_ratio; /* 4, 2 */
} cwm_conj_config;

typedef struct cwm_conj_model cwm_conj_model;
CWM_API int cwm_conj_create(const cwm_conj_config* cfg, cwm_conj_model** out);
CWM_API void cwm_conj_destroy(cwm_conj_model* m);
CWM_API int cwm_conj_load_weight(cwm_conj_model* m, const char* key, const float* data, int on_device, const int64_t* shape, int ndim);
CWM_API int cwm_conj_missing_weights(cwm_conj_model* m, char* buf, int buflen);

typedef struct cwm_conj_forward_args {
    uint32_t struct_size;        /* sizeof(cwm_conj_forward_args) as the caller knows it (see cwm_forward_args): y_ctx_tokens_dev, added in 0.5, is read only
                                  * when the size covers it */
    const float* x_dev;          /* frames, strides as in cwm_forward_args */
    int64_t x_stride_b, x_stride_c, x_stride_t;
    int32_t normalize;
    const uint8_t* mask_dev;     /* bool [B, Nt]; rows MAY have different visible counts (null-token padding) */
    int32_t batch;
    int32_t n_vis_max;           /* max over rows of the visible count (max - min must be <= main_max_pad) */
    const float* ctx_dev;        /* IMU [B, ctx_in_chans, ctx_seq_len] contiguous fp32 */
    const uint8_t* ctx_mask_dev; /* bool [B, ctx_seq_len / ctx_tubelet] */
    int32_t n_vis_ctx_max;
    float* y_tokens_dev;         /* out [B, Nt + main_max_pad - n_vis_max, in_chans*patch*patch]; rows at masked pad slots are 0 */
    int32_t mode;
    int32_t check;
    void* stream;
    /* optional (NULL: not computed): the context stream's predictions, out [B, ctx tokens + ctx_max_pad - n_vis_ctx_max, ctx_in_chans*ctx_tubelet],
     * rows at masked pad slots are 0 -- `forward(..., output_context=True)` returns it, alone or in a tuple with y_tokens
     * (conjoined_vmae.py:852-887, 990-1011) */
    float* y_ctx_tokens_dev;
} cwm_conj_forward_args;

/* replaces: `self.predictor(self._preprocess(x), mask, x_context=..., mask_context=...)` (prediction.py:419-422) */
CWM_API int cwm_conj_forward(cwm_conj_model* m, const cwm_conj_forward_args* args);
CWM_API int cwm_conj_set_lanes(cwm_conj_model* m, int lanes); /* 1 or 2 (anything else: CWM_ERR_INVALID) -- as cwm_model_set_lanes (here the halves must keep >= 12000 encoder rows: batch >= 8); the halves keep the call's n_vis_max / n_vis_ctx_max */
CWM_API int cwm_conj_set_option(cwm_conj_model* m, const char* key, int value); /* as cwm_model_set_option */
CWM_API int cwm_conj_timing_enable(cwm_conj_model* m, int kclass, int enable);
CWM_API int cwm_conj_timing_collect(cwm_conj_model* m, int kclass, struct cwm_kernel_stats* out);

/* ---- timing hooks (bench.py roofline): HIP events around every launch of one kernel class ------ */
#define CWM_KCLASS_GEMM 0        /* every GEMM launch */
#define CWM_KCLASS_ATTENTION 1
#define CWM_KCLASS_GEMM_WIDE 2   /* the GEMM launches that ran the 256x256 8-phase kernel (wide N with at least half a round of tiles: qkv, fc1, whole rounds of fc2) */
#define CWM_KCLASS_GEMM_NARROW 3 /* ... the 128x128 kernels (proj, narrow N, head, patch embed, mixed-tiling remainders); both enabled / collected with class 0 */
/* HBM-bound edge kernels (SURVEY.md 8d "reported separately as achieved GB/s"): for these classes `total_flops` holds the
 * algorithmic BYTES of the launches (each input element read once, each output element written once), not FLOPs */
#define CWM_KCLASS_LAYERNORM 4    /* rows * D * (4 read + 2 * planes written) */
#define CWM_KCLASS_PATCH_GATHER 5 /* visible tokens * C*P*P * (4 read + 2 * planes written) */
#define CWM_KCLASS_FILL_MASK 6    /* B * Nm * Dd * 4 written (the positional rows come from L2) */
#define CWM_KCLASS_UNEMBED 7      /* B * T*C*H*W * (4 read + 4 written) */
/* IMU-conditioned model only (cwm_conj_timing_*): FLOPs again */
#define CWM_KCLASS_CROSS_ATTN 8   /* BidirectionalCrossAttention, both directions: 2 * (4 * N * M * hd) per head */
#define CWM_KCLASS_SMALL_ATTN 9   /* the context stream's self-attention (25..50 tokens) */
#define CWM_KCLASS_COUNT 10
typedef struct cwm_kernel_stats {
    int64_t launches;
    double total_ms;    /* sum of launch durations (HIP events on the launch stream) */
    double total_flops; /* algorithmic FLOPs (2*M*N*K; attention 4*N*N*hd per head) of those launches; BYTES for classes 4-7 */
} cwm_kernel_stats;
CWM_API int cwm_timing_enable(cwm_model* m, int kclass, int enable);
/* Synchronises the recorded events, accumulates, and resets the event pool. */
CWM_API int cwm_timing_collect(cwm_model* m, int kclass, cwm_kernel_stats* out);

/* ---- stand-alone kernel entry points (kernel-level parity tests; same kernels the model uses) -- */
/* fp32 -> bf16 hi (and lo if lo_dev != NULL) */
CWM_API int cwm_split_bf16(const float* x_dev, int64_t n, void* hi_dev, void* lo_dev, void* stream);
/* C[M,N] (fp32, ldc=N) = A[M,K] * W[N,K]^T + bias (+ resid), all fp32 device inputs; the library
 * splits/pads the operands exactly as the model path does.  replaces: F.linear (VideoMAE/utils.py:48-53) */
CWM_API int cwm_linear(const float* a_dev, const float* w_dev, const float* bias_dev, const float* resid_dev, float* c_dev,
               int M, int N, int K, int gelu, int mode, void* stream);
/* O[B,N,H*64] = softmax(q k^T) v per head from a packed qkv activation [B,N,3*H*64] (fp32 device),
 * q scaled by 64^-0.5 after bias as in VideoMAE/utils.py:94-113. */
CWM_API int cwm_attention(const float* qkv_dev, float* o_dev, int B, int N, int H, int mode, void* stream);
/* y = LayerNorm(x) (fp32 in/out, eps, affine).  replaces nn.LayerNorm at VideoMAE/utils.py:148-149 */
CWM_API int cwm_layernorm(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* y_dev, int rows, int D,
                  float eps, void* stream);
/* perm[B,Nt] = [visible ascending | masked ascending]; returns CWM_ERR_MASK if a row's visible
 * count != n_vis (synchronises).  replaces the boolean gathers vmae.py:167,555-556 */
CWM_API int cwm_mask_to_perm(const uint8_t* mask_dev, int B, int Nt, int n_vis, int32_t* perm_dev, void* stream);
/* replaces pred_patches_to_video (prediction.py:245-259); x is [B,T,C,H,W] contiguous raw frames */
CWM_API int cwm_unembed(const float* y_tokens_dev, const float* x_dev, const uint8_t* mask_dev, int B, int T, int C, int H, int W,
                int P, int n_vis, float* out_dev, void* stream);

/* `RectangularizeMasks` (masking.py:90-132) on device masks without reading them back.  The reference equalises the masked count of
 * the rows of a batch by un-masking / masking `torch.randperm(#candidates)[:surplus]` of a row's masked / visible tokens (one draw of the global CPU
 * generator per changed row, in row order).  The draws depend on the per-row COUNTS alone, so:
 *   cwm_mask_row_counts  counts_dev[b] = number of masked (non-zero) tokens of row b                       (the host reads back 4 B per row)
 *   cwm_mask_flip_picks  applies the host's picks in place.  table_dev (int32): [R | rows[R] | offsets[R+1] | to_value[R] | picks[offsets[R]]] --
 *                        for changed row rows[i], every pick k in picks[offsets[i] .. offsets[i+1]) sets the k-th token (ascending position, 0-based,
 *                        counted on the row as it was BEFORE the call) whose state is not to_value[i] (0 = un-mask a masked token, 1 = mask a
 *                        visible one) to to_value[i].  n_rows = R.  Rows of at most 16384 tokens.  Both asynchronous on `stream`. */
CWM_API int cwm_mask_row_counts(const uint8_t* mask_dev, int B, int Nt, int32_t* counts_dev, void* stream);
CWM_API int cwm_mask_flip_picks(uint8_t* mask_dev, int B, int Nt, const int32_t* table_dev, int n_rows, void* stream);

/* Motion-counterfactual prompt construction for B*S prompts at once.
 * replaces: the per-sample loop of FlowGenerator.create_motion_counterfactuals (segmentation.py:324-338)
 *           = PatchPerturbation.forward + ShiftPatchesAndMask.perturb (perturbation.py:99-113, 245-289)
 *           on make_static_movie(x) when fix_passive=1 (prediction.py:731-739), or on MakeStatic(x, masks)
 *           when fix_passive=2 (perturbation.py:120-145 as called at prediction.py:802-803: the patches `masks`
 *           leaves visible take their frame-0 pixels), BEFORE the final mask_rectangularizer call (host).  x [B,T,C,H,W]; active/masks [B*S,Nt] bool ('(b s)' order,
 *           0 = active patch / 0 = passive visible patch); shifts [B*S,2] (dy,dx) in patch units;
 *           outputs x_out [B*S,T,C,H,W], mask_out [B*S,Nt]; either one may be NULL (masks only: what rank 0 of the sharded loop needs
 *           for all prompts before the rectangulariser; frames only: the rows a rank predicts -- x_dev may be NULL with x_out_dev).
 *           Asynchronous on `stream`. */
/* The prompt table of a counterfactual batch -> the dense operands of cwm_shift_prompts, in one launch (no counterpart in the reference, which builds the
 * masks prompt by prompt on the host: interface.py:370-377, segmentation.py:324-338).  table_dev int32 [S,4] rows (active_h, active_w, dy, dx): ONE active
 * patch of frame `frame` (>= 1) per prompt, moved by (dy, dx) patches, nothing passive.  Writes passive [S,Nt] (frame 0 visible, later frames masked),
 * active [S,Nt] (= passive with the prompt's patch cleared) and shifts [S,2].  Cells outside the grid clear nothing.  Asynchronous on `stream`. */
CWM_API int cwm_prompt_table_expand(const int32_t* table_dev, int S, int T, int grid_h, int grid_w, int frame, uint8_t* active_dev, uint8_t* passive_dev,
                            int32_t* shifts_dev, void* stream);
CWM_API int cwm_shift_prompts(const float* x_dev, int B, int T, int C, int H, int W, int P, int frame, int S, int fix_passive,
                      const uint8_t* active_dev, const uint8_t* masks_dev, const int32_t* shifts_dev, float* x_out_dev,
                      uint8_t* mask_out_dev, void* stream);

/* ---- flow-sample statistics: the reductions over the S counterfactual flow samples -------------
 * flows: fp32 device tensor addressed as flows[b*strides[0] + c*strides[1] + y*strides[2] + x*strides[3] + s*strides[4]]
 * (elements), so both the reference's [B,C,H,W,S] view and the sample-major [(b s),C,H,W] batch the flow model emits work
 * without a copy.  All asynchronous on `stream`.
 *
 * cwm_flow_features   x[B][P][S], P = (H/ds)(W/ds): ds x ds average pool per channel, then sqrt(mean_c(.^2))
 *                     replaces: the `_ds` + `distance_func` (= ChannelMSE(dim=1) against zeros) prologue of
 *                     FlowGenerator.compute_flow_corrs (segmentation.py:503-513; utils.py:510-513)
 * cwm_flow_cov        out[B][nrows][P] = rows [row0, row0+nrows) of torch.cov(x[b]) (use_covariance) or torch.corrcoef(x[b])
 *                     over the S samples, NaN -> 0.  replaces: segmentation.py:538-546.  Work buffers: xc [B][P][S], inv_std [B][P].
 *                     Row slabs let ranks that all-gathered `x` each produce a part of the [P,P] matrix (dist.py).
 * cwm_flow_motion_sum sum[B][H*W] = sum_s |flow| (magnitude over channels), each sample first range-normalised over (H,W)
 *                     with max(range, eps) if normalize_per_sample (work buffer: cwm_flow_motion_work_bytes(B, S) bytes, 16-byte aligned).
 *                     replaces: compute_flow_samples_magnitude + the `.mean(-1)` numerator (segmentation.py:250-255, :264-268)
 * cwm_flow_map_finish map = map*scale (scale = 1/S_total), then (map - min)/max(max - min, eps) per b if normalize.
 *                     replaces: segmentation.py:273-275.  Split from the sum so that sample shards can be all-reduced in between. */
CWM_API int cwm_flow_features(const float* flows_dev, const int64_t* strides, int B, int C, int H, int W, int S, int downsample,
                      float* x_dev, void* stream);
/* cwm_flow_transform  the optional prologues of compute_flow_corrs on the pooled features, in place, in the reference's order
 *                     (segmentation.py:519-538; x[b] is its [P, S] matrix): spearman: every row -> argsort over its samples (as floats);
 *                     thresh_mode 1: x * (x > thresh), 2: (x > thresh), 3: ((x - min_P) > thresh * (max_P - min_P)) ("range_thresh");
 *                     normalize: x / max(max_P, eps); zscore: (x - mean_P) / max(std_P, eps), statistics over the P positions per sample.
 *                     stats_work_dev: cwm_flow_transform_work_bytes(B, P, S) bytes, 16-byte aligned (needed by mode 3, normalize, zscore): the [B][S] column
 *                     statistics and the per-chunk partials of the one-pass column reduction */
CWM_API size_t cwm_flow_transform_work_bytes(int B, int P, int S);
CWM_API int cwm_flow_transform(float* x_dev, int B, int P, int S, int spearman, int thresh_mode, float thresh, int normalize, int zscore, float eps,
                       float* stats_work_dev, void* stream);
CWM_API int cwm_flow_cov(const float* x_dev, int B, int P, int S, int row0, int nrows, int use_covariance, float* xc_work_dev,
                 float* inv_std_work_dev, float* out_dev, void* stream);
CWM_API size_t cwm_flow_motion_work_bytes(int B, int S);
CWM_API int cwm_flow_motion_sum(const float* flows_dev, const int64_t* strides, int B, int C, int H, int W, int S,
                        int normalize_per_sample, float eps, float* minmax_work_dev, float* sum_dev, void* stream);
CWM_API int cwm_flow_map_finish(float* map_dev, int B, int HW, float scale, int normalize, float eps, void* stream);

/* ---- collectives around the sharded counterfactual-sampling loop (SURVEY.md 8e / 8b "comm"; BASELINE configs[3]) ----------
 * The reference has no multi-GPU code: it chunks the S prompts of one frame pair over ONE device (prediction.py:513-540,
 * segmentation.py:423-430).  Here the prompts are sharded over one process per GPU and these entry points are the only
 * communication: RCCL (over xGMI) called directly, asynchronous on `stream`, byte counts, device pointers.
 *   cwm_comm_unique_id   rank 0: a fresh 128-byte id, to be handed to the other ranks by any out-of-band means
 *                        (counterfactualworldmodels_amd/dist.py uses the torch.distributed store of the launcher)
 *   cwm_comm_init        collective over all ranks; binds the communicator to the CURRENT HIP device
 *   cwm_broadcast        in place, from `root`: the packed {frame pair | prompt table | masks} buffer
 *   cwm_allgather        equal blocks: recv[r*bytes_per_rank ...] = rank r's send block
 *   cwm_allgatherv       blocks of counts[r] bytes at offsets[r] of recv (host arrays of length nranks); send may alias its own slot
 *   cwm_allreduce_sum_f32  in place (sample-sharded motion map, segmentation.py:257-276)
 * cwm_comm_load(path) optionally names the RCCL shared object to bind (default: the copy already mapped into the process --
 * PyTorch's -- else the ROCm install); cwm_comm_version() returns its NCCL_VERSION_CODE or -1. */
#define CWM_COMM_ID_BYTES 128
typedef struct cwm_comm cwm_comm;
CWM_API int cwm_comm_load(const char* rccl_path);
CWM_API int cwm_comm_version(void);
CWM_API int cwm_comm_unique_id(uint8_t* id_out);
CWM_API int cwm_comm_init(int rank, int nranks, const uint8_t* id_in, cwm_comm** out);
CWM_API void cwm_comm_destroy(cwm_comm* c);
CWM_API int cwm_comm_rank(const cwm_comm* c);
CWM_API int cwm_comm_size(const cwm_comm* c);
CWM_API int cwm_broadcast(cwm_comm* c, void* buf_dev, size_t bytes, int root, void* stream);
CWM_API int cwm_allgather(cwm_comm* c, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream);
CWM_API int cwm_allgatherv(cwm_comm* c, const void* send_dev, void* recv_dev, const size_t* offsets, const size_t* counts, void* stream);
CWM_API int cwm_allreduce_sum_f32(cwm_comm* c, float* buf_dev, size_t count, void* stream);

CWM_API const char* cwm_last_error(void);
/* "cwm_hip <version> gfx950" */
CWM_API const char* cwm_version(void);
/* Hash of the sources this library was built from (counterfactualworldmodels_amd/build.py: source_hash); the Python
 * binding compares it at load time so that a stale in-tree .so is rebuilt instead of silently bound. */
CWM_API const char* cwm_source_hash(void);
/* `HIP version ...; AMD clang version ...` of the hipcc that compiled this library.  The kernels carry hand-counted s_waitcnt instructions whose
 * correctness depends on the ISA the compiler emits around them; tools/asm_lds_lint.py checks that ISA and records the compiler it passed on
 * (csrc/LINT_PASSED.json), build.py and bench.py compare the two. */
CWM_API const char* cwm_compiler_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CWM_HIP_H */
