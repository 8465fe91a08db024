/*
 * cwm_hip_dev.h -- the development entry points of libcwm_hip_dev.so (built by `python -m counterfactualworldmodels_amd.build --dev`):
 * every object of libcwm_hip.so plus csrc/dev.hip.  For tools/ and for the tests that cross-check kernel variants bit for bit; the production
 * library exports none of this.
 */
#ifndef CWM_HIP_DEV_H
#define CWM_HIP_DEV_H

#include "cwm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Tuning hook (tools/autotune_step.py): fix the output-tile configuration of every GEMM launch of one shape -- M, N, K as launched (K padded to
 * 64), epi 0 fp32 / 1 bf16+GELU / 2 bf16 / 3 QKV scatter, overlapped = inside a two-lane forward -- to cfg 1 (128x128), 4 (256x256 8-phase) or 6 (4 for
 * the whole rounds + 1 for the remaining rows); cfg 0 removes the entry, M <= 0 clears the table.  Every configuration gives bit-identical results.
 * The table is process-wide; it reaches the launches of this thread's stand-alone calls and of models created on this thread AFTER the first call. */
CWM_API int cwm_gemm_tile_override(int M, int N, int K, int epi, int overlapped, int cfg);

/* ---- diagnostics: single-kernel micro-benchmarks on random operands (tools/microbench.py) ---------
 * epi: 0 = fp32 out + bias + in-place residual (proj/fc2 form), 1 = bias + GELU -> bf16 (fc1 form),
 *      3 = QKV head scatter (N must be 3*64*heads, M = batch*n_tok with n_tok = M / batch).
 * Runs `iters` back-to-back launches after 3 warm-up launches and returns the mean launch time. */
CWM_API int cwm_bench_gemm(int M, int N, int K, int mode, int epi, int iters, double* avg_us);
CWM_API int cwm_bench_attention(int B, int H, int N, int mode, int iters, double* avg_us);
/* Sets one execution option (the keys of cwm_model_set_option, cwm_hip.h) in THIS THREAD's copy of the options: the stand-alone entry points
 * (cwm_linear, cwm_attention, cwm_bench_* ...) called on the thread afterwards use it, and model handles created on the thread afterwards start from it.
 * A model that already exists is changed with cwm_model_set_option / cwm_conj_set_option.  Also the profiling queries "attn_prof" / "gemm_prof"
 * (per-workgroup timers of builds with -DCWM_ATTN_PROF / -DCWM_GEMM_PROF), and
 * "pretend_device" = d: this thread's wrong-device checks (cwm_forward, cwm_conj_forward, cwm_*_load_weight, the collectives) see device d as current instead of
 * hipGetDevice's answer (-1: off) -- how the refusal is tested on a one-GPU box. */
CWM_API int cwm_debug_set(const char* key, int value);
/* The value of an option in this thread's copy (what a handle created on this thread now would start from). */
CWM_API int cwm_debug_get(const char* key, int* value);


#ifdef __cplusplus
}
#endif
#endif /* CWM_HIP_DEV_H */
