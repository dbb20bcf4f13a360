// Is the rocprofv3 SIGSEGV past ~10 000 graph-launched kernels (profiles/README.md) the tool's, or the engine's capture?
// A 400-node graph of EMPTY kernels -- no engine code at all --, replayed REPLAYS times, to be run under
//     rocprofv3 --kernel-trace --stats -d <dir> -- ./graph_trace_repro 40       (16 000 kernels)
// If this crashes the same way, the limit is the profiler's; if it does not, the engine's captured epoch graph is the suspect.
//     hipcc --offload-arch=gfx950 -O2 -o build_abl/graph_trace_repro scripts/ablation/graph_trace_repro.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void empty_kernel(int* p, int v) { if (p && threadIdx.x == 9999) *p = v; }

int main(int argc, char** argv) {
    const int replays = argc > 1 ? atoi(argv[1]) : 40, nodes = argc > 2 ? atoi(argv[2]) : 400;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, st, (int*)nullptr, i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < replays; ++r) {
        CK(hipGraphLaunch(ge, st));
        if (r % 10 == 9) { CK(hipStreamSynchronize(st)); printf("replay %d done (%d kernels)\n", r + 1, (r + 1) * nodes); fflush(stdout); }
    }
    CK(hipStreamSynchronize(st));
    printf("OK: %d replays of a %d-node graph = %d graph-launched kernels\n", replays, nodes, replays * nodes);
    return 0;
}
