// What does an in-launch seam cost on 32 workgroups of ONE XCD, with hand-offs that stay in that XCD's L2, against the
// placement-independent write-through form?  ("one XCD, L2-scope hand-offs".)
//
// Standalone micro-benchmark (not part of the library):
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build_abl/seam_cost scripts/ablation/seam_cost.hip && build_abl/seam_cost
//
// 256 workgroups of 256 threads are launched; each reads its XCC id (HW_REG_XCC_ID).  In the one-XCD modes only the workgroups that
// landed on XCD 0 take part (they draw a ticket 0 .. P-1; the others leave at once); in the all-XCD mode every workgroup does.
// A "step" models one optimizer step of the training chain: THREE all-to-all seams, at each of which every participant publishes a
// 4 KiB slab and then reads 256 B of every other participant's slab (the transposition of ownership a training step has: K-slice
// partials -> tails, delta1 -> gradient owners, parameters -> everybody).  No arithmetic: the step time IS the seams.
//   mode 0  write-through, any placement (what MI355X_MICROARCH.md prescribes): sc1 stores, s_waitcnt vmcnt(0), workgroup barrier,
//           one agent-scope atomic add on the seam's counter; consumers poll with sc1 loads and read the slabs with sc1 loads
//   mode 1  L2-scope, valid ONLY if all participants share an XCD: plain stores, vmcnt(0), barrier, workgroup-scope atomic add (performed
//           in the XCD's L2); poll and slab reads with nt loads (bypass L1, served by that L2)
// Every spin is bounded (a timeout raises an abort word); the result slab contents are checked on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define TIMEOUT 50000000ull        /* 0.5 s of the 100 MHz clock */
#define SLAB 1024                  /* floats per participant and seam parity (4 KiB) */
#define MAXP 256

struct Ctl {
    unsigned ticket;               // participants so far
    unsigned abort;
    unsigned xcc_count[8];
    unsigned pad[54];
    unsigned seam[3 * 2][32];      // counters: [seam * 2 + parity], each on a line of its own; monotonic per launch
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int MODE>
__global__ __launch_bounds__(256) void seam_kernel(Ctl* ctl, float* slabs, unsigned long long* times, int steps, int want_xcc, int P_expected,
                                                   float* check) {
    __shared__ unsigned s_me, s_ok;
    const int tid = threadIdx.x;
    const unsigned xcc = xcc_id();
    if (tid == 0) {
        atomicAdd(&ctl->xcc_count[xcc & 7], 1u);
        s_me = (want_xcc < 0 || (int)xcc == want_xcc) ? atomicAdd(&ctl->ticket, 1u) : 0xffffffffu;
    }
    __syncthreads();
    const unsigned me = s_me;
    if (me == 0xffffffffu || (int)me >= P_expected) return;      // not on the chosen XCD (or a surplus workgroup): leave
    const int P = P_expected;
    unsigned long long t_begin = 0;
    float acc = 0.f;
    for (int st = 0; st < steps; ++st) {
        if (st == 2 && tid == 0) t_begin = __builtin_amdgcn_s_memrealtime();      // two warm-up steps
        for (int seam = 0; seam < 3; ++seam) {
            const int par = st & 1;
            float* mine = slabs + ((size_t)(seam * 2 + par) * MAXP + me) * SLAB;
            const float val = (float)((st * 3 + seam) * 256 + (int)me);          // an integer < 2^24: exact in float
            // ---- publish: 4 KiB per participant
#pragma unroll
            for (int i = 0; i < SLAB / 256; ++i) {
                float* p = mine + i * 256 + tid;
                if (MODE == 0) __hip_atomic_store(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *p = val;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned* cnt = &ctl->seam[seam * 2 + par][0];
            if (tid == 0) {
                if (MODE == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                // ---- wait until all P have published this (seam, step): the counter of this parity has seen P * (st / 2 + 1) adds
                const unsigned want = (unsigned)P * (unsigned)(st / 2 + 1);
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                unsigned ok = 1;
                for (;;) {
                    unsigned v;
                    if (MODE == 0) v = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else v = __builtin_nontemporal_load(cnt);
                    if (v >= want) break;
                    if (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
                    if (__builtin_amdgcn_s_memrealtime() - t0 > TIMEOUT) { __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                s_ok = ok;
            }
            __syncthreads();
            if (!s_ok) return;
            // ---- consume: 256 B (64 floats) of every participant's slab; thread t reads float (t % 64) of participants t / 64, t / 64 + 4, ...
            for (int q = tid >> 6; q < P; q += 4) {
                const float* theirs = slabs + ((size_t)(seam * 2 + par) * MAXP + q) * SLAB + (tid & 63);
                float v;
                if (MODE == 0) v = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else v = __builtin_nontemporal_load(theirs);
                acc += fabsf(v - (float)((st * 3 + seam) * 256 + q));                   // 0 if the hand-off delivered the fresh value
            }
        }
    }
    if (tid == 0) times[me] = __builtin_amdgcn_s_memrealtime() - t_begin;
    // sum of deviations over the whole run: must be exactly 0
    atomicAdd(check, fabsf(acc));
}

template <int MODE>
static void run(const char* name, int want_xcc, int P, int steps) {
    Ctl* ctl; float* slabs; unsigned long long* times; float* check;
    CK(hipMalloc(&ctl, sizeof(Ctl))); CK(hipMemset(ctl, 0, sizeof(Ctl)));
    CK(hipMalloc(&slabs, (size_t)6 * MAXP * SLAB * 4)); CK(hipMemset(slabs, 0, (size_t)6 * MAXP * SLAB * 4));
    CK(hipMalloc(&times, MAXP * 8)); CK(hipMemset(times, 0, MAXP * 8));
    CK(hipMalloc(&check, 4)); CK(hipMemset(check, 0, 4));
    hipLaunchKernelGGL(seam_kernel<MODE>, dim3(256), dim3(256), 0, 0, ctl, slabs, times, steps, want_xcc, P, check);
    CK(hipDeviceSynchronize());
    Ctl h; std::vector<unsigned long long> t(MAXP); float c;
    CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    CK(hipMemcpy(t.data(), times, MAXP * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&c, check, 4, hipMemcpyDeviceToHost));
    unsigned long long tmax = 0;
    for (int i = 0; i < P; ++i) tmax = t[i] > tmax ? t[i] : tmax;
    printf("%-58s P = %3d  workgroups per XCD %u %u %u %u %u %u %u %u  participants %u  abort %u  stale-sum %g  ->  %.2f us per step = %.2f us per seam\n",
           name, P, h.xcc_count[0], h.xcc_count[1], h.xcc_count[2], h.xcc_count[3], h.xcc_count[4], h.xcc_count[5], h.xcc_count[6], h.xcc_count[7],
           h.ticket, h.abort, c, tmax / 100.0 / (steps - 2), tmax / 100.0 / (steps - 2) / 3.0);
    fflush(stdout);
    CK(hipFree(ctl)); CK(hipFree(slabs)); CK(hipFree(times)); CK(hipFree(check));
}

int main() {
    const int steps = 202;
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("write-through (sc1), 32 workgroups of XCD 0", 0, 32, steps);
        run<1>("L2-scope (plain stores, nt loads), 32 workgroups of XCD 0", 0, 32, steps);
        run<0>("write-through (sc1), 27 workgroups of XCD 0", 0, 27, steps);
        run<1>("L2-scope, 27 workgroups of XCD 0", 0, 27, steps);
        run<0>("write-through (sc1), 32 workgroups on ANY XCD", -1, 32, steps);
        run<0>("write-through (sc1), all 256 workgroups", -1, 256, steps);
    }
    return 0;
}
