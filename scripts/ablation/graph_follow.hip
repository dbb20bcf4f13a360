// Does work queued BEHIND a running hipGraph slow the graph down?  (The engine's deferred loss collection -- the next iteration's forward
// enqueued while the epoch's graph executes -- measured 2 % / 7 % slower at 40 / 400 graph kernels: 0.5 us per graph kernel.)
//   A  graph of n dependent ~10 us kernels, nothing behind it
//   B  the same, a long eager kernel enqueued on the SAME stream right after hipGraphLaunch
//   C  the long kernel on a SECOND stream, gated by an event recorded behind the graph
//   D  the same n kernels launched eagerly (no graph), nothing behind
//   E  eager, the long kernel enqueued behind them at once
// Reported: duration of the n kernels (HIP events around them), median of 21 repetitions.
//   hipcc --offload-arch=gfx950 -O2 scripts/ablation/graph_follow.hip -o build_abl/graph_follow && build_abl/graph_follow 40 400
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin_kernel(unsigned long long ticks, unsigned* sink) {        // ticks of s_memrealtime (100 MHz)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
    if (threadIdx.x == 0 && blockIdx.x == 0 && sink) atomicAdd(sink, 1u);
}

static float median(std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv) {
    unsigned* sink; CK(hipMalloc(&sink, 4)); CK(hipMemset(sink, 0, 4));
    hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1, eg; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&eg, hipEventDisableTiming));
    for (int a = 1; a < (argc > 1 ? argc : 2); ++a) {
        const int n = argc > 1 ? atoi(argv[a]) : 40;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s, 1000ull, sink);     // ~10 us each
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        const char* names[5] = {"A graph, nothing behind", "B graph, long kernel behind on the same stream", "C graph, long kernel on a second stream behind an event",
                                "D eager, nothing behind", "E eager, long kernel behind"};
        for (int mode = 0; mode < 5; ++mode) {
            std::vector<float> t;
            for (int rep = 0; rep < 21; ++rep) {
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, s));
                if (mode <= 2) CK(hipGraphLaunch(ge, s));
                else for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, s, 1000ull, sink);
                CK(hipEventRecord(e1, s));
                if (mode == 1 || mode == 4) hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(512), 0, s, 30000ull, sink);         // 300 us
                if (mode == 2) { CK(hipEventRecord(eg, s)); CK(hipStreamWaitEvent(s2, eg, 0)); hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(512), 0, s2, 30000ull, sink); }
                CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                t.push_back(ms * 1000.f);
            }
            printf("n = %4d  %-62s %9.1f us  (%.2f us per kernel)\n", n, names[mode], median(t), median(t) / n);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
