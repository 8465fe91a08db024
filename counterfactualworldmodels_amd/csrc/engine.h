// Shared host-side machinery of the model handles behind the C ABI: packed weights, state-dict slots,
// activation workspace, timed launches and the transformer Block sequence.  Used by model.hip (plain
// VMAE predictor) and conj_model.hip (IMU-conditioned conjoined predictor).
#pragma once
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/cwm_hip.h"
#include "common.h"
#include "kernels.h"

namespace cwm {

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct LinearW {
    bf16* w = nullptr;     // fast mode:   [Npad][Kpad]   bf16(weight)
    bf16* w_il = nullptr;  // parity mode: [Npad][2*Kpad] hi/lo interleaved per 32-k block (common.h a_pos)
    int64_t plane = 0;
    int N = 0, K = 0, Npad = 0, Kpad = 0;
    float* bias = nullptr;  // [Npad] (zero-filled) or nullptr when the layer has no bias
};

struct BlockW {
    float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
    LinearW qkv, proj, fc1, fc2;
};

enum SlotKind { SLOT_MATRIX, SLOT_VECTOR, SLOT_IGNORED };

struct Slot {
    SlotKind kind = SLOT_VECTOR;
    std::vector<int64_t> shape;
    LinearW* lin = nullptr;  // SLOT_MATRIX
    float* dst = nullptr;    // SLOT_VECTOR (device)
    int repeat = 1;          // SLOT_VECTOR: number of consecutive copies written at dst
    int64_t numel = 0;
    bool loaded = false;
};

struct EventPair {
    hipEvent_t a, b;
    double flops;
    int sub;  // kernel class the launch is also booked under (0: none)
};

struct KernelTimer {
    bool enabled = false;
    std::vector<EventPair> pool;
    size_t used = 0;
    cwm_kernel_stats acc = {0, 0.0, 0.0};
};

// Activation buffers of one token stream (bf16 planes are [2][rows][width], plane stride set per use).
struct StreamBuffers {
    bf16 *hbuf = nullptr, *gbuf = nullptr, *qbuf = nullptr, *kbuf = nullptr, *vbuf = nullptr;
    float* qkv_f32 = nullptr;  // only for streams whose head_dim != 64 (small-sequence attention path)
};

// Encoder rows (batch elements x visible tokens) each half of a batch must keep for the forward to run as two batch lanes (cwm_forward,
// cwm_conj_forward).  Measured (tools/lane_threshold.py, round 4): ViT-B/8 batch 8 / 10 / 14 (3168 / 3960 / 5544 rows per half) -6 / -7.5 / -10 % with
// two lanes, batch 6 (2376) +3 %; ViT-L/4 batch 2 (3168) -7 %.  (Rounds 1-3 used 6000: batch >= 16.)
constexpr int kMinLaneRows = 3000;  // (Tuning.min_lane_rows overrides it per model: tools/lane_threshold.py)
// The IMU-conditioned model (its lanes carry a context stream each: four queues): batch 2 / 3 / 4 / 6 are 5.3 / 4.3 / 2.5 / 2.8 % SLOWER on two lanes,
// batch 8 / 12 / 16 2.3 / 1.4 / 1.0 % faster (3172 visible rows per sample; rounds 1-3 split from batch 4)
constexpr int kMinLaneRowsConj = 12000;

struct Engine {
    int device = 0;
    int overlapped = 0;  // 1 while a forward call runs two batch lanes (passed to the GEMM kernel choice)
    Tuning tune = thread_tuning();  // this model's execution options (cwm_model_set_option); every launch of the model carries a pointer to it
    float ln_eps = 1e-6f;
    std::map<std::string, Slot> slots;
    std::vector<void*> allocs;     // weights etc., freed on destroy
    std::vector<void*> ws_allocs;  // workspace, re-allocated when it has to grow
    KernelTimer timers[CWM_KCLASS_COUNT];
    struct SplitKWs {
        float* slabs;
        unsigned* counts;
        uint64_t last_use;  // launch counter at the last split launch on the stream: the eviction order
    };
    uint64_t splitk_clock = 0;
    static constexpr size_t kMaxSplitKStreams = 8;
    std::map<hipStream_t, SplitKWs> splitk_ws;  // one split-K workspace per stream that has launched a split GEMM (batch lanes run concurrently); created lazily, capped

    ~Engine();
    int alloc(void** p, size_t bytes, bool zero, bool workspace);
    template <typename T>
    int ws(T** p, size_t count) {
        void* v = nullptr;
        if (int rc = alloc(&v, count * sizeof(T), true, true)) return rc;
        *p = (T*)v;
        return 0;
    }
    int free_workspace();

    int make_linear(LinearW& L, int N, int K, bool bias);
    int make_vec(float** v, int n);
    void add_matrix_slot(const std::string& key, LinearW* L, std::vector<int64_t> shape);
    void add_vec_slot(const std::string& key, float* dst, std::vector<int64_t> shape, int repeat = 1);
    void add_ignored_slot(const std::string& key, std::vector<int64_t> shape);
    int make_block(BlockW& b, const std::string& pre, int D, int hidden);
    int make_sinusoid(float** dst, int n_pos, int d, int extra_rows = 0);      // VideoMAE/utils.py:251-268 (float64 host)
    int make_pos_embedding_f32(float** dst, int n_pos, int d, int extra_rows = 0);  // transformer.py:37-52 (float32)

    int load_weight(const char* key, const float* data, int on_device, const int64_t* shape, int ndim);
    int missing_weights(char* buf, int buflen);

    int run_gemm(const GemmParams& p, int planes, hipStream_t s);
    int run_attention(const AttnParams& p, int planes, hipStream_t s);
    // the HBM-bound edge kernels, booked by class with their algorithmic bytes (cwm_hip.h CWM_KCLASS_*)
    int run_layernorm(const LayerNormParams& p, int planes, hipStream_t s);
    int run_patch_gather(const PatchGatherParams& p, int planes, hipStream_t s);
    int run_index_gather(const PatchGatherParams& p, const uint8_t* mask, int n_vis, int* perm, int* rank, int* err_rows, int planes, hipStream_t s);
    int run_fill_mask_tokens(float* x_full, const float* mask_token, const float* pos, const int* perm, int B, int Nt, int n_vis, int D, hipStream_t s);
    int run_unembed(const UnembedParams& p, hipStream_t s);
    // `launch()` between an event pair of class `kclass` (no events unless the class is enabled); work = FLOPs or bytes
    template <typename F>
    int timed(int kclass, double work, hipStream_t s, F&& launch) {
        EventPair* e = nullptr;
        if (int rc = timer_begin(kclass, work, s, &e)) return rc;
        if (int rc = launch()) return rc;
        return timer_end(e, s);
    }
    int timer_begin(int kclass, double work, hipStream_t s, EventPair** out);
    int timer_end(EventPair* e, hipStream_t s);
    // Block.forward (VideoMAE/utils.py:146-153) on a residual stream x[B*n_tok, D] (in place), head_dim 64
    int run_block(const BlockW& w, float* x, int B, int n_tok, int D, int H, int planes, StreamBuffers& sb, hipStream_t s, int n_keep = 0);
    // same for short sequences with any head_dim (fp32 VALU attention): the IMU context stream
    int run_block_small(const BlockW& w, float* x, int B, int n_tok, int D, int H, int planes, StreamBuffers& sb, hipStream_t s);

    int timing_enable(int kclass, int enable);
    int timing_collect(int kclass, cwm_kernel_stats* out);
};

GemmParams gemm_base(const bf16* A, int lda, const LinearW& L, int M, int planes);

// fp32 [N][K] -> bf16 [Npad][Kpad] (hi only) and [Npad][2*Kpad] (hi/lo interleaved), zero padded
int launch_pack_weight(const float* src, int N, int K, bf16* hi, bf16* il, int Npad, int Kpad, hipStream_t stream);

}  // namespace cwm
