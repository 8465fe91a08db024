// Development library (libcwm_hip_dev.so = every object of libcwm_hip.so + this file; include/cwm_hip_dev.h): the switches, per-shape tile
// overrides, micro-benchmarks on random operands and profiling queries that tools/ and the bitwise cross-checks of the test suite use.  None of
// it is linked into the production library.  cwm_debug_set changes THIS THREAD's copy of the execution options (kernels.h thread_tuning):
// the stand-alone entry points called on the thread afterwards use it, model handles created on the thread afterwards start from it; a model
// that already exists is changed through cwm_model_set_option / cwm_conj_set_option, which the production library has too.
#include <atomic>
#include <map>
#include <mutex>
#include <tuple>

#include "../../include/cwm_hip_dev.h"
#include "engine.h"

using namespace cwm;

void cwm_set_pretend_device(int d);  // engine.hip

namespace {
struct Scratch {
    std::vector<void*> ptrs;
    ~Scratch() {
        for (void* p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    T* get(size_t count, bool zero = false) {
        void* p = nullptr;
        if (hipMalloc(&p, count * sizeof(T) + 16) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        if (zero) (void)hipMemset(p, 0, count * sizeof(T));
        return (T*)p;
    }
};
}  // namespace

namespace {
__global__ void fill_random_bf16_kernel(bf16* dst, int64_t n, unsigned seed, float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    dst[i] = (bf16)(((float)(x & 0xFFFF) / 32768.0f - 1.0f) * scale);
}
__global__ void fill_random_f32_kernel(float* dst, int64_t n, unsigned seed, float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    dst[i] = ((float)(x & 0xFFFF) / 32768.0f - 1.0f) * scale;
}
void fill_bf16(bf16* d, int64_t n, unsigned seed, float scale) {
    hipLaunchKernelGGL(fill_random_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, n, seed, scale);
}
void fill_f32(float* d, int64_t n, unsigned seed, float scale) {
    hipLaunchKernelGGL(fill_random_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, n, seed, scale);
}
}  // namespace


extern "C" int cwm_debug_set(const char* key, int value) {
    CWM_REQUIRE(key, "cwm_debug_set: null key");
    if (!strcmp(key, "gemm_prof")) return gemm_prof_dump();            // query (profiling builds)
    if (!strcmp(key, "attn_prof")) return attention_pipe_prof(value);  // query (profiling builds)
    if (!strcmp(key, "pretend_device")) {  // the device index this thread's wrong-device checks see (-1: the real one): tests on a one-GPU box
        cwm_set_pretend_device(value);
        return CWM_OK;
    }
    CWM_REQUIRE(tuning_set(thread_tuning(), key, value) == 0, "cwm_debug_set: unknown key %s", key);
    return CWM_OK;
}

extern "C" int cwm_debug_get(const char* key, int* value) {
    CWM_REQUIRE(key && value, "cwm_debug_get: null argument");
    CWM_REQUIRE(tuning_get(thread_tuning(), key, value) == 0, "cwm_debug_get: unknown key %s", key);
    return CWM_OK;
}

// Per-shape overrides of the tile choice (the tuning hook behind tools/autotune_step.py).  All configurations give bit-identical results
// (tests/test_kernels_gpu.py), so an override can only change the speed.  The table is process-wide; it reaches a launch through
// Tuning.tile_hook, which the first override installs in this thread's options (models created afterwards inherit it).
namespace {
struct TileKey {
    int M, N, K, epi, ovl;
    bool operator<(const TileKey& o) const { return std::tie(M, N, K, epi, ovl) < std::tie(o.M, o.N, o.K, o.epi, o.ovl); }
};
std::mutex g_tile_mu;
std::map<TileKey, int> g_tile_overrides;
std::atomic<int> g_tile_override_count{0};
int tile_hook(int M, int N, int K, int epi, int overlapped) {
    if (g_tile_override_count.load(std::memory_order_relaxed) == 0) return 0;
    std::lock_guard<std::mutex> lock(g_tile_mu);
    auto it = g_tile_overrides.find(TileKey{M, N, K, epi, overlapped ? 1 : 0});
    return it == g_tile_overrides.end() ? 0 : it->second;
}
}  // namespace

extern "C" int cwm_gemm_tile_override(int M, int N, int K, int epi, int overlapped, int cfg) {
    thread_tuning().tile_hook = tile_hook;
    std::lock_guard<std::mutex> lock(g_tile_mu);
    if (M <= 0) {
        g_tile_overrides.clear();
    } else if (cfg == 0) {
        g_tile_overrides.erase(TileKey{M, N, K, epi, overlapped ? 1 : 0});
    } else {
        CWM_REQUIRE(cfg == 1 || cfg == 4 || cfg == 6, "cwm_gemm_tile_override: unknown tile configuration %d", cfg);
        g_tile_overrides[TileKey{M, N, K, epi, overlapped ? 1 : 0}] = cfg;
    }
    g_tile_override_count.store((int)g_tile_overrides.size());
    return 0;
}

extern "C" int cwm_bench_gemm(int M, int N, int K, int mode, int epi, int iters, double* avg_us) {
    CWM_REQUIRE(avg_us && M > 0 && N > 0 && K > 0 && iters > 0, "cwm_bench_gemm: bad argument");
    CWM_REQUIRE(mode == CWM_MODE_FAST || mode == CWM_MODE_PARITY, "cwm_bench_gemm: bad mode");
    const int planes = mode == CWM_MODE_PARITY ? 2 : 1;
    const int Kp = round_up(K, 64), Np = round_up(N, 256);
    Scratch sc;
    bf16* A = sc.get<bf16>((size_t)2 * M * Kp);
    bf16* W = sc.get<bf16>((size_t)2 * Np * Kp);
    float* bias = sc.get<float>(Np);
    float* Cm = sc.get<float>((size_t)M * N);
    bf16* G = sc.get<bf16>((size_t)2 * M * N + 64 * 1024);
    bf16* G2 = sc.get<bf16>((size_t)2 * M * N + 64 * 1024);
    bf16* G3 = sc.get<bf16>((size_t)2 * M * N + 64 * 1024 * 64);
    CWM_REQUIRE(A && W && bias && Cm && G && G2 && G3, "cwm_bench_gemm: out of device memory");
    fill_bf16(A, (int64_t)2 * M * Kp, 1, 1.0f);
    fill_bf16(W, (int64_t)2 * Np * Kp, 2, 0.05f);
    fill_f32(bias, Np, 3, 0.1f);
    fill_f32(Cm, (int64_t)M * N, 4, 1.0f);
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = Kp; p.W = W;
    p.M = M; p.N = N; p.K = Kp; p.bias = bias;
    if (epi == 1 || epi == 2) {
        p.epi = epi == 1 ? EPI_BF16_GELU : EPI_BF16; p.out_hi = G; p.ldo = N;
    } else if (epi == 3) {
        CWM_REQUIRE(N % 192 == 0, "cwm_bench_gemm: QKV epilogue needs N = 3*64*heads");
        const int D = N / 3, H = D / 64, n_tok = 792 <= M && M % 792 == 0 ? 792 : M, B = M / n_tok;
        (void)B;
        p.epi = EPI_QKV; p.rows_in = n_tok; p.rows_out = n_tok; p.map_stride = n_tok;
        p.q_out = G; p.k_out = G2; p.v_out = G3; p.qk_plane = (int64_t)M * D;
        p.qkv_dim = D; p.heads = H; p.head_dim = 64; p.n_tok = n_tok; p.q_scale = 0.125f;
    } else {
        p.epi = EPI_F32; p.C = Cm; p.ldc = N; p.resid = Cm; p.ldr = N;
    }
    p.tune = &thread_tuning();
    hipEvent_t e0, e1;
    CWM_HIP_CHECK(hipEventCreate(&e0));
    CWM_HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        if (int rc = launch_gemm(p, planes, 0)) return rc;
    CWM_HIP_CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i)
        if (int rc = launch_gemm(p, planes, 0)) return rc;
    CWM_HIP_CHECK(hipEventRecord(e1, 0));
    CWM_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CWM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_us = 1e3 * ms / iters;
    return CWM_OK;
}

extern "C" int cwm_bench_attention(int B, int H, int N, int mode, int iters, double* avg_us) {
    CWM_REQUIRE(avg_us && B > 0 && H > 0 && N > 0 && iters > 0, "cwm_bench_attention: bad argument");
    CWM_REQUIRE(mode == CWM_MODE_FAST || mode == CWM_MODE_PARITY, "cwm_bench_attention: bad mode");
    const int planes = mode == CWM_MODE_PARITY ? 2 : 1;
    const int D = H * 64;
    const int64_t qk_plane = (int64_t)B * N * D;
    Scratch sc;
    bf16* q = sc.get<bf16>(2 * qk_plane);
    bf16* k = sc.get<bf16>(2 * qk_plane);
    bf16* v = sc.get<bf16>(2 * qk_plane);
    bf16* o = sc.get<bf16>(2 * qk_plane);
    CWM_REQUIRE(q && k && v && o, "cwm_bench_attention: out of device memory");
    fill_bf16(q, 2 * qk_plane, 5, 0.5f);
    fill_bf16(k, 2 * qk_plane, 6, 1.0f);
    fill_bf16(v, 2 * qk_plane, 7, 1.0f);
    AttnParams a;
    memset(&a, 0, sizeof(a));
    a.q = q; a.k = k; a.v = v; a.qk_plane = qk_plane; a.o = o; a.o_plane = qk_plane; a.ldo = D;
    a.n_tok = N; a.heads = H; a.batch = B;
    a.tune = &thread_tuning();
    hipEvent_t e0, e1;
    CWM_HIP_CHECK(hipEventCreate(&e0));
    CWM_HIP_CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        if (int rc = launch_attention(a, planes, 0)) return rc;
    CWM_HIP_CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i)
        if (int rc = launch_attention(a, planes, 0)) return rc;
    CWM_HIP_CHECK(hipEventRecord(e1, 0));
    CWM_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CWM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_us = 1e3 * ms / iters;
    return CWM_OK;
}
