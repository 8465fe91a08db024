// Back-to-back dependent kernel launches on one stream: stream launches vs a captured hipGraph (MI355X).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_gap tools/launch_gap.hip && /tmp/launch_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void spin_kernel(float* p, int iters) {
    float v = p[threadIdx.x & 63];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    if (v == 123.f) p[0] = v;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    float* d;
    (void)hipMalloc(&d, 1 << 20);
    (void)hipMemset(d, 0, 1 << 20);
    hipStream_t s;
    (void)hipStreamCreate(&s);
    const int n = 136;
    for (int blocks : {1, 2048}) {
        for (int iters : {0, 20000}) {
            // plain stream launches
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipStreamSynchronize(s);
                const double t0 = now_us();
                for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, s, d, iters);
                const double t1 = now_us();
                (void)hipStreamSynchronize(s);
                const double t2 = now_us();
                if (rep == 2) printf("blocks %5d iters %6d  stream: enqueue %.2f us/launch, total %.2f us/kernel\n", blocks, iters, (t1 - t0) / n, (t2 - t0) / n);
            }
            // graph
            hipGraph_t g;
            hipGraphExec_t ge;
            (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, s, d, iters);
            (void)hipStreamEndCapture(s, &g);
            (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipStreamSynchronize(s);
                const double t0 = now_us();
                (void)hipGraphLaunch(ge, s);
                (void)hipStreamSynchronize(s);
                const double t2 = now_us();
                if (rep == 2) printf("blocks %5d iters %6d  graph : total %.2f us/kernel\n", blocks, iters, (t2 - t0) / n);
            }
            (void)hipGraphExecDestroy(ge);
            (void)hipGraphDestroy(g);
        }
    }
    return 0;
}
