// What would a forward FUSED FROM COORDINATES cost at villin shape?  
//
// Standalone ablation harness (not part of the library; results are not computed correctly on purpose -- timing only):
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_out/fused_floor scripts/ablation/fused_floor.hip && gpurun_out/fused_floor
//
// The fused design the brief describes: ONE 16-row unit per CU (its coordinates, 16 x 591 atoms in 16-byte cells = 148 KiB, fill
// the LDS), eight waves split the 54 feature chunks, every lane builds its own MFMA A operands (8 distances = 16 LDS gathers,
// LayerNorm sums, f16 hi / lo split) and the layer-1 weight planes (2 x f16, fragment-linear: 995 328 B) are streamed from L2
// straight into registers -- ONCE PER UNIT, because a CU holds only 16 rows.  The terms, each switchable:
//     W  : the weight-plane stream (18 KiB per chunk and wave from the L2-resident image)
//     M  : the 27 v_mfma_f32_16x16x32_f16 per chunk
//     C  : the unit's coordinates HBM -> registers (next unit, prefetched) -> LDS cells
//     G  : gathers (ds_read_b128, conflict-free [atom][row] cells) + distance + sums + split per lane
// The existing two-kernel design stages the same planes once per 256 rows (16 units per CU and pass).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int A = 591, NCH = 54, NT = 9, ROWF = 3 * A;       // villin: 591 atoms, Fp = 1728 = 54 chunks, 144 hidden = 9 tiles
constexpr int CELLS = A * 16;                                  // [atom][row] cells of 16 B
constexpr int APT = (CELLS + 511) / 512;                       // atoms per thread and unit (19)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
__device__ f32x3 raw_buffer_load_f32x3(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v3f32");
__device__ u32x4 raw_buffer_load_u32x4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
__device__ __forceinline__ i32x4 rsrc_of(const void* p) {
    const unsigned long long a = (unsigned long long)p;
    return (i32x4){(int)(unsigned)a, (int)(unsigned)(a >> 32), 0x7fffffff, 0x00020000};
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool W, bool M, bool C, bool G>
__global__ __launch_bounds__(512) void fused_floor_kernel(const float* __restrict__ coords, const u32x4* __restrict__ Wimg,
                                                          const unsigned* __restrict__ pairs, float* __restrict__ out, int units) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    // this wave's chunks: wave, wave + 8, ... (7 for waves 0..5, 6 for waves 6, 7)
    constexpr int MAXC = (NCH + 7) / 8;
    // pair table (54 x 32 pairs, a << 16 | b in atom units) in the LDS tail behind the cells: 8 pairs per lane and chunk = two
    // ds_read_b128 (in registers for the whole launch it would be 56 of a wave's 256)
    unsigned* ptab = (unsigned*)(lds + CELLS * 4);
    if (G) {
        for (int i = tid; i < NCH * 32; i += 512) ptab[i] = pairs[i];
    }
    f32x3 nx[APT];                                              // the NEXT unit's atoms of this thread
    int goff[APT], loff[APT];                                   // per-thread constants: byte offset in the unit's rows / LDS cell
#pragma unroll
    for (int i = 0; i < APT; ++i) {
        int cell = i * 512 + tid;                               // fetch order: a row's atoms contiguous (768 B per wave-instruction)
        if (cell > CELLS - 1) cell = CELLS - 1;
        const int row = cell / A, atom = cell - row * A;
        goff[i] = (row * ROWF + atom * 3) * 4;
        loff[i] = (atom * 16 + row) * 4;
    }
    auto fetch = [&](int u) {
        if (!C) return;
        const i32x4 r = rsrc_of(coords + (long long)u * 16 * ROWF);
#pragma unroll
        for (int i = 0; i < APT; ++i) nx[i] = raw_buffer_load_f32x3(r, goff[i], 0, 2);
    };
    auto deposit = [&]() {
        if (!C) return;
#pragma unroll
        for (int i = 0; i < APT; ++i) *(f32x4*)(lds + loff[i]) = (f32x4){nx[i][0], nx[i][1], nx[i][2], 0.f};
    };
    int u = blockIdx.x;
    if (u < units) fetch(u);
    float check = 0.f;
    for (; u < units; u += gridDim.x) {
        __syncthreads();                                        // every wave is done with the previous unit's cells
        deposit();
        __syncthreads();
        if (u + (int)gridDim.x < units) fetch(u + gridDim.x);   // in flight under this unit's whole K loop
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float s1 = 0.f, s2 = 0.f;
        // weight fragments: ring of two tile GROUPS (3 tiles x 2 planes = 6 KiB per wave), one group ahead of the MFMAs: 48 KiB
        // in flight per CU (64 B/clk x ~700 clk of loaded L2 latency)
        constexpr int TG = 3, NG = NT / TG;
        u32x4 wf[2][2 * TG];
        const i32x4 rw = rsrc_of(Wimg);
        auto wload = [&](int slot, int c, int g) {
#pragma unroll
            for (int i = 0; i < TG; ++i) {
                wf[slot][i] = raw_buffer_load_u32x4(rw, lane * 16, ((c * 2 * NT + g * TG + i) * 64) * 16, 0);
                wf[slot][TG + i] = raw_buffer_load_u32x4(rw, lane * 16, ((c * 2 * NT + NT + g * TG + i) * 64) * 16, 0);
            }
        };
        if (W) wload(0, wave, 0);
        const int nmine = (NCH - wave + 7) / 8;                 // 7 (waves 0..5) or 6
#pragma unroll 1
        for (int k2 = 0; k2 < MAXC + 1; k2 += 2) {              // two chunks per trip: the ring parity is the same at every trip
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int k = k2 + kk;
                if (k < nmine) {                                // wave-uniform
                    const int c = wave + 8 * k;
                    const int cn = k + 1 < nmine ? c + 8 : c;
                    f16x8 hi, lo;
                    if (G) {
                        const u32x4 p0 = *(const u32x4*)(ptab + c * 32 + lq * 8), p1 = *(const u32x4*)(ptab + c * 32 + lq * 8 + 4);
                        float d[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const unsigned p = j < 4 ? p0[j & 3] : p1[j & 3];
                            const f32x4 pa = *(const f32x4*)(lds + ((p >> 16) * 16 + li) * 4);
                            const f32x4 pb = *(const f32x4*)(lds + ((p & 0xffffu) * 16 + li) * 4);
                            const float dx = pa[0] - pb[0], dy = pa[1] - pb[1], dz = pa[2] - pb[2];
                            d[j] = __builtin_sqrtf(fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
                            s1 += d[j];
                            s2 = fmaf(d[j], d[j], s2);
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float v = d[j] * 4096.f - 6144.f;      // fixed shift and power-of-two scale (stand-in)
                            const _Float16 h = (_Float16)v;
                            hi[j] = h;
                            lo[j] = (_Float16)(v - (float)h);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { hi[j] = (_Float16)(float)(lane + j + k); lo[j] = (_Float16)0.25f; }
                    }
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const int slot = (kk * NG + g) & 1;
                        __builtin_amdgcn_sched_barrier(0);
                        if (W) {
                            if (g + 1 < NG) wload(slot ^ 1, c, g + 1);
                            else wload(slot ^ 1, cn, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < TG; ++i) {
                            const int t = g * TG + i;
                            f16x8 wh, wl;
                            if (W) {
                                wh = __builtin_bit_cast(f16x8, wf[slot][i]);
                                wl = __builtin_bit_cast(f16x8, wf[slot][TG + i]);
                            } else {
#pragma unroll
                                for (int j = 0; j < 8; ++j) { wh[j] = (_Float16)(float)(t + 1); wl[j] = (_Float16)0.5f; }
                            }
                            if (M) {
                                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo, wh, acc[t], 0, 0, 0);
                                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, wl, acc[t], 0, 0, 0);
                                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, wh, acc[t], 0, 0, 0);
                            } else {
                                asm volatile("" :: "v"(wh), "v"(wl), "v"(hi), "v"(lo));
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) check += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
        check += s1 + s2;
    }
    if (check == 12345.678f) out[blockIdx.x * 512 + tid] = check;   // keeps the work alive, (practically) never stores
}

template <bool W, bool M, bool C, bool G>
static float run(const char* name, const float* coords, const u32x4* Wimg, const unsigned* pairs, float* out, int units, int reps) {
    const size_t ldsb = C || G ? (size_t)CELLS * 16 + NCH * 32 * 4 : 1024;
    CK(hipFuncSetAttribute((const void*)fused_floor_kernel<W, M, C, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fused_floor_kernel<W, M, C, G>), dim3(256), dim3(512), ldsb, 0, coords, Wimg, pairs, out, units);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((fused_floor_kernel<W, M, C, G>), dim3(256), dim3(512), ldsb, 0, coords, Wimg, pairs, out, units);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-34s %8.3f ms per launch  (%d units of 16 rows; %.2f us per unit and CU)\n", name, ms, units, ms * 1e3 / ((units + 255) / 256));
    fflush(stdout);
    return ms;
}

int main() {
    const int rows = 210000, units = rows / 16;                 // C3: N K + N rows
    float* coords; u32x4* Wimg; unsigned* pairs; float* out;
    CK(hipMalloc(&coords, (size_t)rows * ROWF * 4));
    CK(hipMalloc(&Wimg, (size_t)NCH * 2 * NT * 64 * 16));
    CK(hipMalloc(&pairs, NCH * 32 * 4));
    CK(hipMalloc(&out, 256 * 512 * 4));
    {
        std::vector<float> h((size_t)rows * ROWF);
        unsigned s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) * (3.0f / 16777216.f); }
        CK(hipMemcpy(coords, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        std::vector<unsigned short> w((size_t)NCH * 2 * NT * 64 * 8);
        for (auto& v : w) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3000 + ((s >> 12) & 0x7ff)); }     // f16 in [0.125, 0.25)
        CK(hipMemcpy(Wimg, w.data(), w.size() * 2, hipMemcpyHostToDevice));
        std::vector<unsigned> p(NCH * 32);
        for (auto& v : p) {
            s = s * 1664525u + 1013904223u; const unsigned a = (s >> 8) % A;
            s = s * 1664525u + 1013904223u; const unsigned b = (s >> 8) % A;
            v = a << 16 | b;
        }
        CK(hipMemcpy(pairs, p.data(), p.size() * 4, hipMemcpyHostToDevice));
    }
    printf("fused-from-coordinates forward, villin shape: %d rows, %d atoms, %d chunks x %d tiles, weight planes %d bytes per unit\n",
           rows, A, NCH, NT, NCH * 2 * NT * 64 * 16);
    const int reps = 20;
    run<true, false, false, false>("W   (weight stream alone)", coords, Wimg, pairs, out, units, reps);
    run<false, true, false, false>("M   (MFMAs alone)", coords, Wimg, pairs, out, units, reps);
    run<true, true, false, false>("W+M", coords, Wimg, pairs, out, units, reps);
    run<false, false, true, false>("C   (coordinate stream alone)", coords, Wimg, pairs, out, units, reps);
    run<false, false, true, true>("C+G (coordinates, gathers, split)", coords, Wimg, pairs, out, units, reps);
    run<false, true, true, true>("C+G+M", coords, Wimg, pairs, out, units, reps);
    run<true, true, true, false>("W+M+C", coords, Wimg, pairs, out, units, reps);
    run<true, true, true, true>("W+M+C+G (the whole K loop)", coords, Wimg, pairs, out, units, reps);
    return 0;
}
