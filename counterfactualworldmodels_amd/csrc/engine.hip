// Engine: weight packing, state-dict slots, workspace, timed launches, transformer Block sequence.
#include <stdarg.h>

#include "engine.h"

using namespace cwm;

// ---------------------------------------------------------------------------------------------
// error plumbing (C ABI never throws)
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

void cwm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* cwm_last_error(void) { return g_err; }

#include <mutex>
#include <set>
int cwm_set_max_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    CWM_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, kernel})) return 0;
    CWM_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({dev, kernel});
    return 0;
}
// (development library only -- csrc/dev.hip "pretend_device": the device index this THREAD's checks see instead of hipGetDevice's, so that the
// wrong-device refusal can be exercised on a one-GPU box; -1 = off.  libcwm_hip.so exports no way to set it.)
static thread_local int g_pretend_device = -1;
void cwm_set_pretend_device(int d) { g_pretend_device = d; }
int cwm_require_device(int handle_device, const char* what) {
    int cur = -1;
    CWM_HIP_CHECK(hipGetDevice(&cur));
    if (g_pretend_device >= 0) cur = g_pretend_device;
    CWM_REQUIRE(cur == handle_device, "%s: the handle was created on HIP device %d but the calling thread's current device is %d (hipSetDevice(%d) first)", what,
                handle_device, cur, handle_device);
    return 0;
}
extern "C" const char* cwm_version(void) { return "cwm_hip 0.7.0 gfx950"; }
#ifndef CWM_SRC_HASH
#define CWM_SRC_HASH "unknown"
#endif
extern "C" const char* cwm_source_hash(void) { return CWM_SRC_HASH; }
#ifndef CWM_HIPCC_VERSION
#define CWM_HIPCC_VERSION "unknown"
#endif
extern "C" const char* cwm_compiler_version(void) { return CWM_HIPCC_VERSION; }

namespace cwm {

// ---- execution options -------------------------------------------------------------------------
const Tuning& default_tuning() {
    static const Tuning t;
    return t;
}
Tuning& thread_tuning() {
    static thread_local Tuning t;
    return t;
}
namespace {
struct TuningField { const char* name; int Tuning::*member; };
}
static const TuningField* tuning_field(const char* key) {
    typedef TuningField Field;
    static const Field fields[] = {{"gemm_tile", &Tuning::gemm_tile}, {"gemm_debug", &Tuning::gemm_debug}, {"gemm_staged", &Tuning::gemm_staged},
                                   {"gemm_direct", &Tuning::gemm_direct}, {"attn_kernel", &Tuning::attn_kernel}, {"attn_remap", &Tuning::attn_remap},
                                   {"attn_tail", &Tuning::attn_tail}, {"attn_ksplit", &Tuning::attn_ksplit}, {"prune_last_block", &Tuning::prune_last_block}, {"index_fused", &Tuning::index_fused},
                                   {"min_lane_rows", &Tuning::min_lane_rows}, {"conj_ctx_stream", &Tuning::conj_ctx_stream}, {"conj_attn", &Tuning::conj_attn}};
    for (const Field& f : fields)
        if (!strcmp(key, f.name)) return &f;
    return nullptr;
}
int tuning_set(Tuning& t, const char* key, int value) {
    const TuningField* f = tuning_field(key);
    if (!f) return -1;
    t.*(f->member) = value;
    return 0;
}
// What cwm_model_set_option / cwm_conj_set_option accept: every option, minus the values that make a forward return WRONG outputs with CWM_OK -- the
// timing-only ablation bits of "gemm_debug" (1 skip the epilogue's stores, 2 skip the epilogue, 8 skip every LayerNorm).  Those stay reachable through
// the development library alone (cwm_debug_set on the thread that then creates the model).  0, -1 unknown key, -2 development-only value.
int tuning_set_production(Tuning& t, const char* key, int value) {
    if (!strcmp(key, "gemm_debug") && (value & (1 | 2 | 8))) return -2;
    return tuning_set(t, key, value);
}
int tuning_get(const Tuning& t, const char* key, int* value) {
    const TuningField* f = tuning_field(key);
    if (!f) return -1;
    *value = t.*(f->member);
    return 0;
}

__global__ void pack_weight_kernel(const float* src, int N, int K, bf16* hi, bf16* il, int Npad, int Kpad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Npad * Kpad) return;
    const int n = (int)(i / Kpad), k = (int)(i - (int64_t)n * Kpad);
    float v = (n < N && k < K) ? src[(size_t)n * K + k] : 0.f;
    bf16 h, l;
    split_bf16(v, h, l);
    if (hi) hi[i] = h;
    if (il) {
        bf16* d = il + a_pos<2>(n, Kpad, k);
        d[0] = h;
        d[kLoOffset] = l;
    }
}

int launch_pack_weight(const float* src, int N, int K, bf16* hi, bf16* il, int Npad, int Kpad, hipStream_t stream) {
    const int64_t total = (int64_t)Npad * Kpad;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, N, K, hi, il, Npad, Kpad);
    CWM_HIP_CHECK(hipGetLastError());
    return 0;
}

Engine::~Engine() {
    (void)hipDeviceSynchronize();
    for (void* p : allocs) (void)hipFree(p);
    for (void* p : ws_allocs) (void)hipFree(p);
    for (auto& kv : splitk_ws) {
        (void)hipFree(kv.second.slabs);
        (void)hipFree(kv.second.counts);
    }
    for (auto& t : timers)
        for (auto& e : t.pool) {
            (void)hipEventDestroy(e.a);
            (void)hipEventDestroy(e.b);
        }
}

int Engine::alloc(void** p, size_t bytes, bool zero, bool workspace) {
    CWM_HIP_CHECK(hipMalloc(p, bytes ? bytes : 16));
    (workspace ? ws_allocs : allocs).push_back(*p);
    if (zero) CWM_HIP_CHECK(hipMemset(*p, 0, bytes ? bytes : 16));
    return 0;
}

int Engine::free_workspace() {
    CWM_HIP_CHECK(hipDeviceSynchronize());
    for (void* p : ws_allocs) (void)hipFree(p);
    ws_allocs.clear();
    return 0;
}

int Engine::make_linear(LinearW& L, int N, int K, bool bias) {
    L.N = N;
    L.K = K;
    L.Npad = round_up(N, 256);
    L.Kpad = round_up(K, 64);
    L.plane = (int64_t)L.Npad * L.Kpad;
    void* p;
    if (int rc = alloc(&p, (size_t)L.plane * sizeof(bf16), true, false)) return rc;
    L.w = (bf16*)p;
    if (int rc = alloc(&p, (size_t)2 * L.plane * sizeof(bf16), true, false)) return rc;
    L.w_il = (bf16*)p;
    if (bias) {
        if (int rc = alloc(&p, (size_t)L.Npad * sizeof(float), true, false)) return rc;
        L.bias = (float*)p;
    }
    return 0;
}

int Engine::make_vec(float** v, int n) {
    void* p;
    if (int rc = alloc(&p, (size_t)n * sizeof(float), true, false)) return rc;
    *v = (float*)p;
    return 0;
}

void Engine::add_matrix_slot(const std::string& key, LinearW* L, std::vector<int64_t> shape) {
    Slot s;
    s.kind = SLOT_MATRIX;
    s.shape = shape;
    s.lin = L;
    s.numel = (int64_t)L->N * L->K;
    slots[key] = s;
}

void Engine::add_vec_slot(const std::string& key, float* dst, std::vector<int64_t> shape, int repeat) {
    Slot s;
    s.kind = SLOT_VECTOR;
    s.shape = shape;
    s.dst = dst;
    s.repeat = repeat;
    s.numel = 1;
    for (auto d : shape) s.numel *= d;
    slots[key] = s;
}

void Engine::add_ignored_slot(const std::string& key, std::vector<int64_t> shape) {
    Slot s;
    s.kind = SLOT_IGNORED;
    s.shape = shape;
    s.numel = 1;
    for (auto d : shape) s.numel *= d;
    slots[key] = s;
}

int Engine::make_block(BlockW& b, const std::string& pre, int D, int hidden) {
    int rc;
    if ((rc = make_vec(&b.ln1_g, D)) || (rc = make_vec(&b.ln1_b, D)) || (rc = make_vec(&b.ln2_g, D)) || (rc = make_vec(&b.ln2_b, D)))
        return rc;
    if ((rc = make_linear(b.qkv, 3 * D, D, true)) || (rc = make_linear(b.proj, D, D, true)) || (rc = make_linear(b.fc1, hidden, D, true)) ||
        (rc = make_linear(b.fc2, D, hidden, true)))
        return rc;
    add_vec_slot(pre + "norm1.weight", b.ln1_g, {D});
    add_vec_slot(pre + "norm1.bias", b.ln1_b, {D});
    add_vec_slot(pre + "norm2.weight", b.ln2_g, {D});
    add_vec_slot(pre + "norm2.bias", b.ln2_b, {D});
    // qkv bias = [q_bias | 0 | v_bias]  (VideoMAE/utils.py:89-93: there is no k bias)
    add_vec_slot(pre + "attn.q_bias", b.qkv.bias, {D});
    add_vec_slot(pre + "attn.v_bias", b.qkv.bias + 2 * D, {D});
    add_matrix_slot(pre + "attn.qkv.weight", &b.qkv, {3 * D, D});
    add_matrix_slot(pre + "attn.proj.weight", &b.proj, {D, D});
    add_vec_slot(pre + "attn.proj.bias", b.proj.bias, {D});
    add_matrix_slot(pre + "mlp.fc1.weight", &b.fc1, {hidden, D});
    add_vec_slot(pre + "mlp.fc1.bias", b.fc1.bias, {hidden});
    add_matrix_slot(pre + "mlp.fc2.weight", &b.fc2, {D, hidden});
    add_vec_slot(pre + "mlp.fc2.bias", b.fc2.bias, {D});
    return 0;
}

// `get_sinusoid_encoding_table` (VideoMAE/utils.py:251-268): float64 on the host, cast to fp32.
int Engine::make_sinusoid(float** dst, int n_pos, int d, int extra_rows) {
    std::vector<float> tab((size_t)(n_pos + extra_rows) * d, 0.f);
    for (int pos = 0; pos < n_pos; ++pos)
        for (int j = 0; j < d; ++j) {
            const double ang = (double)pos / pow(10000.0, 2.0 * (double)(j / 2) / (double)d);
            tab[(size_t)pos * d + j] = (float)((j & 1) ? cos(ang) : sin(ang));
        }
    if (int rc = make_vec(dst, (n_pos + extra_rows) * d)) return rc;
    CWM_HIP_CHECK(hipMemcpy(*dst, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

// `pos_embedding` (transformer.py:37-52): the same formula evaluated in float32 (torch ops restated:
// freqs = powf(10000, 2*trunc(j/2)/d) with the division done in float32, angle = pos / freqs).
int Engine::make_pos_embedding_f32(float** dst, int n_pos, int d, int extra_rows) {
    std::vector<float> tab((size_t)(n_pos + extra_rows) * d, 0.f);
    for (int j = 0; j < d; ++j) {
        const float e = (2.0f * (float)(j / 2)) / (float)d;
        const float freq = powf(10000.0f, e);
        for (int pos = 0; pos < n_pos; ++pos) {
            const float ang = (float)pos / freq;
            tab[(size_t)pos * d + j] = (j & 1) ? cosf(ang) : sinf(ang);
        }
    }
    if (int rc = make_vec(dst, (n_pos + extra_rows) * d)) return rc;
    CWM_HIP_CHECK(hipMemcpy(*dst, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int Engine::load_weight(const char* key, const float* data, int on_device, const int64_t* shape, int ndim) {
    CWM_REQUIRE(key && data && shape, "load_weight: null argument");
    if (int rc = cwm_require_device(device, "load_weight")) return rc;
    auto it = slots.find(key);
    CWM_REQUIRE(it != slots.end(), "unexpected key in state_dict: %s", key);
    Slot& s = it->second;
    bool same = (int)s.shape.size() == ndim;
    for (int i = 0; same && i < ndim; ++i) same = s.shape[i] == shape[i];
    CWM_REQUIRE(same, "size mismatch for %s", key);
    if (s.kind == SLOT_IGNORED) {
        s.loaded = true;
        return CWM_OK;
    }
    if (s.kind == SLOT_VECTOR) {
        for (int r = 0; r < s.repeat; ++r)
            CWM_HIP_CHECK(hipMemcpy(s.dst + (size_t)r * s.numel, data, (size_t)s.numel * sizeof(float),
                                    on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    } else {
        LinearW& L = *s.lin;
        const float* src = data;
        float* tmp = nullptr;
        if (!on_device) {
            CWM_HIP_CHECK(hipMalloc((void**)&tmp, (size_t)s.numel * sizeof(float)));
            hipError_t e = hipMemcpy(tmp, data, (size_t)s.numel * sizeof(float), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                (void)hipFree(tmp);
                cwm_set_error("hipMemcpy failed: %s", hipGetErrorString(e));
                return CWM_ERR_HIP;
            }
            src = tmp;
        }
        const int64_t total = (int64_t)L.Npad * L.Kpad;
        hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, src, L.N, L.K, L.w, L.w_il, L.Npad,
                           L.Kpad);
        hipError_t e = hipDeviceSynchronize();
        if (tmp) (void)hipFree(tmp);
        if (e != hipSuccess) {
            cwm_set_error("pack_weight failed: %s", hipGetErrorString(e));
            return CWM_ERR_HIP;
        }
    }
    s.loaded = true;
    return CWM_OK;
}

int Engine::missing_weights(char* buf, int buflen) {
    int missing = 0;
    if (buf && buflen > 0) buf[0] = 0;
    for (auto& kv : slots)
        if (!kv.second.loaded) {
            if (!missing && buf && buflen > 0) snprintf(buf, buflen, "%s", kv.first.c_str());
            ++missing;
        }
    return missing;
}

// ---- timed launches ---------------------------------------------------------------------------
int Engine::timer_begin(int kclass, double flops, hipStream_t s, EventPair** out) {
    KernelTimer& t = timers[kclass];
    *out = nullptr;
    if (!t.enabled) return 0;
    if (t.used == t.pool.size()) {
        EventPair e;
        CWM_HIP_CHECK(hipEventCreate(&e.a));
        CWM_HIP_CHECK(hipEventCreate(&e.b));
        t.pool.push_back(e);
    }
    EventPair& e = t.pool[t.used++];
    e.flops = flops;
    e.sub = 0;
    CWM_HIP_CHECK(hipEventRecord(e.a, s));
    *out = &e;
    return 0;
}

int Engine::timer_end(EventPair* e, hipStream_t s) {
    if (e) CWM_HIP_CHECK(hipEventRecord(e->b, s));
    return 0;
}

int Engine::run_gemm(const GemmParams& p_in, int planes, hipStream_t s) {
    GemmParams p = p_in;
    p.overlapped = overlapped;
    p.tune = &tune;
    const int cfg = gemm_choose_tile(p, planes);
    {   // a launch (or the 128x128 remainder of a mixed-tiling launch: fc2 of a single-lane batch) that takes the deep-ring kernel's split-K
        // path carries this stream's workspace, created on first need (32 MB + counters; at most kMaxSplitKStreams per engine: a caller
        // that rotates streams evicts the oldest)
        GemmParams big, rest;
        const bool split = (cfg == 1 && gemm_splitk_parts(p, planes) > 1) ||
                           (cfg == 6 && gemm_mixed_split(p, &big, &rest) && gemm_splitk_parts(rest, planes) > 1);
        if (split) {
            auto it = splitk_ws.find(s);
            if (it == splitk_ws.end()) {
                if (splitk_ws.size() >= kMaxSplitKStreams) {
                    // evict the workspace of the stream that used one longest ago.  (hipFree waits for the device, so a launch of that stream
                    // that still reads it finishes first; the stream itself may be gone -- a caller that rotates streams has usually destroyed
                    // it --, so it is not touched.)  The entry leaves the table before anything here can fail.
                    auto victim = splitk_ws.begin();
                    for (auto w = splitk_ws.begin(); w != splitk_ws.end(); ++w)
                        if (w->second.last_use < victim->second.last_use) victim = w;
                    const SplitKWs old = victim->second;
                    splitk_ws.erase(victim);
                    (void)hipFree(old.slabs);
                    (void)hipFree(old.counts);
                }
                SplitKWs w = {nullptr, nullptr, 0};
                if (int rc = splitk_workspace_alloc(&w.slabs, &w.counts, s)) return rc;
                it = splitk_ws.emplace(s, w).first;
            }
            it->second.last_use = ++splitk_clock;
            p.sk2_slabs = it->second.slabs;
            p.sk2_count = it->second.counts;
        }
    }
    GemmParams part[2];
    int cfgs[2] = {cfg, 0}, nparts = 1;
    part[0] = p;
    if (cfg == 6 && timers[CWM_KCLASS_GEMM].enabled && gemm_mixed_split(p, &part[0], &part[1])) {
        // timed runs book the two kernels of a mixed-tiling launch separately (so that the per-kernel averages are those rocprofv3 sees)
        cfgs[0] = 4;
        cfgs[1] = 1;
        nparts = 2;
    }
    for (int i = 0; i < nparts; ++i) {
        EventPair* e;
        if (int rc = timer_begin(CWM_KCLASS_GEMM, 2.0 * part[i].M * (double)part[i].N * part[i].K, s, &e)) return rc;
        if (e) e->sub = (cfgs[i] == 3 || cfgs[i] == 4) ? CWM_KCLASS_GEMM_WIDE : CWM_KCLASS_GEMM_NARROW;
        if (int rc = (nparts == 2 ? launch_gemm_tile(part[i], planes, cfgs[i], s) : launch_gemm(part[i], planes, s))) return rc;
        if (int rc = timer_end(e, s)) return rc;
    }
    return 0;
}

int Engine::run_attention(const AttnParams& p_in, int planes, hipStream_t s) {
    AttnParams p = p_in;
    p.tune = &tune;
    EventPair* e;
    const double fl = 4.0 * (double)p.n_tok * p.n_tok * 64.0 * p.heads * p.batch;
    if (int rc = timer_begin(CWM_KCLASS_ATTENTION, fl, s, &e)) return rc;
    if (int rc = launch_attention(p, planes, s)) return rc;
    return timer_end(e, s);
}

int Engine::run_layernorm(const LayerNormParams& p, int planes, hipStream_t s) {
    if (tune.gemm_debug & 8) return 0;  // ablation ("gemm_debug" bit 3): what the step would cost without any LayerNorm launch (timing only)
    return timed(CWM_KCLASS_LAYERNORM, (double)p.rows * p.D * (4.0 + 2.0 * planes), s, [&] { return launch_layernorm(p, planes, s); });
}

int Engine::run_patch_gather(const PatchGatherParams& p, int planes, hipStream_t s) {
    return timed(CWM_KCLASS_PATCH_GATHER, (double)p.B * p.n_rows * p.C * p.P * p.P * (4.0 + 2.0 * planes), s, [&] { return launch_patch_gather(p, planes, s); });
}

int Engine::run_index_gather(const PatchGatherParams& p, const uint8_t* mask, int n_vis, int* perm, int* rank, int* err_rows, int planes, hipStream_t s) {
    // booked with the patch gather's class and bytes (+ the mask row in, the permutation and its inverse out)
    const int L = p.perm_stride ? p.perm_stride : p.Nt;
    return timed(CWM_KCLASS_PATCH_GATHER, (double)p.B * p.n_rows * p.C * p.P * p.P * (4.0 + 2.0 * planes) + (double)p.B * L * (1.0 + 4.0 + (rank ? 4.0 : 0.0)), s,
                 [&] { return launch_index_gather(p, mask, n_vis, perm, rank, err_rows, planes, s); });
}

int Engine::run_fill_mask_tokens(float* x_full, const float* mask_token, const float* pos, const int* perm, int B, int Nt, int n_vis, int D,
                                 hipStream_t s) {
    return timed(CWM_KCLASS_FILL_MASK, (double)B * (Nt - n_vis) * D * 4.0, s,
                 [&] { return launch_fill_mask_tokens(x_full, mask_token, pos, perm, B, Nt, n_vis, D, s); });
}

int Engine::run_unembed(const UnembedParams& p, hipStream_t s) {
    return timed(CWM_KCLASS_UNEMBED, (double)p.B * p.T * p.C * p.H * p.W * 8.0, s, [&] { return launch_unembed(p, s); });
}

int Engine::timing_enable(int kclass, int enable) {
    CWM_REQUIRE(kclass >= 0 && kclass < CWM_KCLASS_COUNT, "timing_enable: bad kernel class");
    KernelTimer& t = timers[kclass];
    t.enabled = enable != 0;
    while (enable && t.pool.size() < 512) {
        EventPair e;
        CWM_HIP_CHECK(hipEventCreate(&e.a));
        CWM_HIP_CHECK(hipEventCreate(&e.b));
        e.flops = 0;
        e.sub = 0;
        t.pool.push_back(e);
    }
    return CWM_OK;
}

int Engine::timing_collect(int kclass, cwm_kernel_stats* out) {
    CWM_REQUIRE(out && kclass >= 0 && kclass < CWM_KCLASS_COUNT, "timing_collect: bad argument");
    KernelTimer& t = timers[kclass];
    for (size_t i = 0; i < t.used; ++i) {
        CWM_HIP_CHECK(hipEventSynchronize(t.pool[i].b));
        float ms = 0.f;
        CWM_HIP_CHECK(hipEventElapsedTime(&ms, t.pool[i].a, t.pool[i].b));
        t.acc.launches += 1;
        t.acc.total_ms += ms;
        t.acc.total_flops += t.pool[i].flops;
        if (t.pool[i].sub > 0 && t.pool[i].sub < CWM_KCLASS_COUNT) {  // per-kernel split of the GEMM class
            cwm_kernel_stats& sa = timers[t.pool[i].sub].acc;
            sa.launches += 1;
            sa.total_ms += ms;
            sa.total_flops += t.pool[i].flops;
        }
    }
    t.used = 0;
    *out = t.acc;
    t.acc = {0, 0.0, 0.0};
    return CWM_OK;
}

GemmParams gemm_base(const bf16* A, int lda, const LinearW& L, int M, int planes) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A;
    p.lda = lda;
    p.W = planes == 2 ? L.w_il : L.w;
    p.M = M;
    p.N = L.N;
    p.K = L.Kpad;
    p.bias = L.bias;
    return p;
}

// Block.forward (VideoMAE/utils.py:146-153): x += proj(attn(LN1 x)); x += fc2(gelu(fc1(LN2 x)))
// n_keep (0 = all): only the LAST n_keep tokens of every sample are needed downstream (the last decoder block: the decoder returns
// head(norm(x[:, -Nm:])), vmae.py:250-251).  Keys / values still come from all tokens; queries, proj, LN2 and the MLP run on the
// kept rows only (compact activations, residual rows addressed with an offset).  Kept rows are bit-identical to the full block.
int Engine::run_block(const BlockW& w, float* x, int B, int n_tok, int D, int H, int planes, StreamBuffers& sb, hipStream_t s, int n_keep) {
    const int M = B * n_tok;
    const int64_t hplane = (int64_t)M * D;
    const int hidden = w.fc1.N;
    const bool part = n_keep > 0 && n_keep < n_tok;
    const int n_out = part ? n_keep : n_tok, Mo = B * n_out, first = n_tok - n_out;
    int rc;
    LayerNormParams ln;
    memset(&ln, 0, sizeof(ln));
    ln.x = x; ln.ldx = D; ln.gamma = w.ln1_g; ln.beta = w.ln1_b; ln.eps = ln_eps; ln.D = D; ln.rows = M;
    ln.out = sb.hbuf; ln.out_plane = hplane; ln.ldo = D;
    if ((rc = run_layernorm(ln, planes, s))) return rc;

    GemmParams g = gemm_base(sb.hbuf, D, w.qkv, M, planes);
    g.epi = EPI_QKV;
    g.rows_in = n_tok; g.rows_out = n_tok; g.map_stride = n_tok;
    g.q_out = sb.qbuf; g.k_out = sb.kbuf; g.v_out = sb.vbuf;
    g.qk_plane = hplane;
    g.qkv_dim = D; g.heads = H; g.head_dim = D / H; g.n_tok = n_tok;
    g.q_scale = 1.0f / sqrtf((float)(D / H));
    if ((rc = run_gemm(g, planes, s))) return rc;

    AttnParams a;
    memset(&a, 0, sizeof(a));
    a.q = sb.qbuf; a.k = sb.kbuf; a.v = sb.vbuf; a.qk_plane = hplane;
    a.o = sb.hbuf; a.o_plane = (int64_t)Mo * D; a.ldo = D; a.n_tok = n_tok; a.heads = H; a.batch = B;
    if (part) { a.q_off = first; a.n_q = n_out; }
    if ((rc = run_attention(a, planes, s))) return rc;

    // the two residual GEMMs
    auto residual = [&](GemmParams& r) {
        r.epi = EPI_F32; r.C = x; r.ldc = D; r.resid = x; r.ldr = D;
        if (part) { r.rows_in = n_out; r.rows_out = n_tok; r.out_row_offset = first; }
    };
    g = gemm_base(sb.hbuf, D, w.proj, Mo, planes);
    residual(g);
    if ((rc = run_gemm(g, planes, s))) return rc;

    ln.gamma = w.ln2_g; ln.beta = w.ln2_b;
    if (part) { ln.rows = Mo; ln.rows_out_per_b = n_out; ln.rows_in_per_b = n_tok; ln.in_offset = first; ln.out_plane = (int64_t)Mo * D; }
    if ((rc = run_layernorm(ln, planes, s))) return rc;
    g = gemm_base(sb.hbuf, D, w.fc1, Mo, planes);
    g.epi = EPI_BF16_GELU; g.out_hi = sb.gbuf; g.out_plane = (int64_t)Mo * hidden; g.ldo = hidden;
    if ((rc = run_gemm(g, planes, s))) return rc;

    g = gemm_base(sb.gbuf, hidden, w.fc2, Mo, planes);
    residual(g);
    return run_gemm(g, planes, s);
}

// The same Block for a short sequence (n_tok <= 64) with any head_dim <= 64: qkv stays fp32 and the
// attention is the small fp32 VALU kernel (the IMU context stream: 25/50 tokens, head_dim 32).
int Engine::run_block_small(const BlockW& w, float* x, int B, int n_tok, int D, int H, int planes, StreamBuffers& sb, hipStream_t s) {
    const int M = B * n_tok;
    const int64_t hplane = (int64_t)M * D;
    const int hidden = w.fc1.N;
    int rc;
    LayerNormParams ln;
    memset(&ln, 0, sizeof(ln));
    ln.x = x; ln.ldx = D; ln.gamma = w.ln1_g; ln.beta = w.ln1_b; ln.eps = ln_eps; ln.D = D; ln.rows = M;
    ln.out = sb.hbuf; ln.out_plane = hplane; ln.ldo = D;
    if ((rc = run_layernorm(ln, planes, s))) return rc;

    GemmParams g = gemm_base(sb.hbuf, D, w.qkv, M, planes);
    g.epi = EPI_F32; g.C = sb.qkv_f32; g.ldc = 3 * D;
    if ((rc = run_gemm(g, planes, s))) return rc;

    SmallAttnParams a;
    memset(&a, 0, sizeof(a));
    a.qkv = sb.qkv_f32; a.B = B; a.n_tok = n_tok; a.heads = H; a.head_dim = D / H;
    a.o = sb.hbuf; a.o_plane = hplane; a.ldo = D;
    if ((rc = timed(CWM_KCLASS_SMALL_ATTN, 4.0 * (double)B * H * n_tok * n_tok * (D / H), s, [&] {
             return (tune.conj_attn && small_attention_mfma_ok(n_tok, D / H) && D % 32 == 0) ? launch_small_attention_mfma(a, planes, s) : launch_small_attention(a, planes, s);
         })))
        return rc;

    g = gemm_base(sb.hbuf, D, w.proj, M, planes);
    g.epi = EPI_F32; g.C = x; g.ldc = D; g.resid = x; g.ldr = D;
    if ((rc = run_gemm(g, planes, s))) return rc;

    ln.gamma = w.ln2_g; ln.beta = w.ln2_b;
    if ((rc = run_layernorm(ln, planes, s))) return rc;

    g = gemm_base(sb.hbuf, D, w.fc1, M, planes);
    g.epi = EPI_BF16_GELU; g.out_hi = sb.gbuf; g.out_plane = (int64_t)M * hidden; g.ldo = hidden;
    if ((rc = run_gemm(g, planes, s))) return rc;

    g = gemm_base(sb.gbuf, hidden, w.fc2, M, planes);
    g.epi = EPI_F32; g.C = x; g.ldc = D; g.resid = x; g.ldr = D;
    return run_gemm(g, planes, s);
}

}  // namespace cwm
