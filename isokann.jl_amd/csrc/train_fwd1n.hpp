// train_fwd1n's device code (see kernels_train.hip for the design notes): shared by the kernel of that name and by the kernel that
// carries the tail step behind it (kernels_train_fwdtail.hip).
#pragma once
#include "isk_internal.h"

typedef float fw1n_f32x4 __attribute__((ext_vector_type(4)));

#define FW1N_WAVES 8
#define FW1N_NTB 5
// SC1: the K-slice partial is handed to ANOTHER workgroup of the same launch -- stored write-through
template <int NTB, bool LN, bool SC1>
__device__ __forceinline__ void fwd1n_body(const Fwd1nArgs& a, float* red) {
    typedef fw1n_f32x4 f32x4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int rb = blockIdx.x;
    const int tg = blockIdx.y % a.ntg, ks = blockIdx.y / a.ntg;
    const int t0 = tg * NTB;                                              // first tile of this group
    int r0 = rb * 32 + li, r1 = r0 + 16;
    if (r0 > a.B - 1) r0 = a.B - 1;
    if (r1 > a.B - 1) r1 = a.B - 1;
    const int x0r = a.perm ? a.perm[r0] : r0, x1r = a.perm ? a.perm[r1] : r1;      // consumed after the first weight loads are out
    const f32x4* Wp = (const f32x4*)a.W1n + lane;
    const int cbeg = ks * a.cpb;
    int cend = cbeg + a.cpb;
    if (cend > a.nch) cend = a.nch;
    f32x4 acc0[NTB], acc1[NTB];
#pragma unroll
    for (int t = 0; t < NTB; ++t) { acc0[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    struct Frag { f32x4 xa, xb, g, b, w[NTB]; };
    const float* x0 = a.X + lq * 4;
    const float* x1 = a.X + lq * 4;
    bool xset = false;
    auto load = [&](Frag& f, int c) {
        // the weight / LayerNorm loads first: the sample addresses depend on the batch's row indices (a load of their own),
        // and a wave issues in order -- whatever stands behind the first sample load waits for that index with it
#pragma unroll
        for (int t = 0; t < NTB; ++t) {
            const int tt = t0 + t < a.nt ? t0 + t : a.nt - 1;             // a short last group re-reads its last tile (discarded)
            f.w[t] = Wp[(long long)(c * a.nt + tt) * 64];
        }
        if (LN) { f.g = *(const f32x4*)(a.g + c * 16 + lq * 4); f.b = *(const f32x4*)(a.beta + c * 16 + lq * 4); }
        if (!xset) { x0 += (long long)x0r * a.Fp; x1 += (long long)x1r * a.Fp; xset = true; }
        f.xa = *(const f32x4*)(x0 + c * 16);
        f.xb = *(const f32x4*)(x1 + c * 16);
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float pa = LN ? fmaf(f.xa[j], f.g[j], f.b[j]) : f.xa[j];       // LayerNorm affine on the operand
            const float pb = LN ? fmaf(f.xb[j], f.g[j], f.b[j]) : f.xb[j];
#pragma unroll
            for (int t = 0; t < NTB; ++t) {
                acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, f.w[t][j], acc0[t], 0, 0, 0);
                acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pb, f.w[t][j], acc1[t], 0, 0, 0);
            }
        }
    };
    // two fragment sets in flight: the next chunk's loads are issued before this chunk's MFMAs (clamped, branch-free)
    Frag fa, fb;
    const int c0 = cbeg + wave;
    const int clast = cend - 1;
    load(fa, c0 < clast ? c0 : clast);
    for (int c = c0; c < cend; c += 2 * FW1N_WAVES) {
        load(fb, c + FW1N_WAVES < clast ? c + FW1N_WAVES : clast);
        __builtin_amdgcn_sched_barrier(0);
        mma(fa);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 2 * FW1N_WAVES < cend) load(fa, c + 2 * FW1N_WAVES);     // wave-uniform; rare (long slices only)
        __builtin_amdgcn_sched_barrier(0);
        if (c + FW1N_WAVES < cend) mma(fb);
        __builtin_amdgcn_sched_barrier(0);
    }
    // two stages keep the buffer at four tile sets (two workgroups per CU fit beside the featurizer's LDS image):
    // waves 4..7 park their tiles, waves 0..3 add them to their own and park the sums
    const int wp = wave & 3;
    if (wave >= 4) {
#pragma unroll
        for (int t = 0; t < NTB; ++t) {
            *(f32x4*)(red + (((wp * 2 + 0) * NTB + t) * 64 + lane) * 4) = acc0[t];
            *(f32x4*)(red + (((wp * 2 + 1) * NTB + t) * 64 + lane) * 4) = acc1[t];
        }
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
        for (int t = 0; t < NTB; ++t) {
            f32x4* p0 = (f32x4*)(red + (((wp * 2 + 0) * NTB + t) * 64 + lane) * 4);
            f32x4* p1 = (f32x4*)(red + (((wp * 2 + 1) * NTB + t) * 64 + lane) * 4);
            *p0 += acc0[t];                                  // same lane reads and rewrites its own slot: no barrier in between
            *p1 += acc1[t];
        }
    }
    __syncthreads();
    // 2*NTB*64 accumulator quads, summed over the four wave pairs and stored: quad q = (unit, tile, lane') -> rows 4 lq' .. + 3, column 16 t + li'
    const int H1p = a.nt * 16;
    float* zp = a.zp + (long long)ks * a.zstride;
    for (int q = threadIdx.x; q < 2 * NTB * 64; q += FW1N_WAVES * 64) {
        f32x4 v = *(const f32x4*)(red + q * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *(const f32x4*)(red + (w * 2 * NTB * 64 + q) * 4);
        const int u = q / (NTB * 64), rem = q - u * NTB * 64, t = rem >> 6, lp = rem & 63;
        if (t0 + t >= a.nt) continue;
        const long long row = (long long)(rb * 2 + u) * 16 + (lp >> 4) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* p = zp + (row + r) * H1p + (t0 + t) * 16 + (lp & 15);
            if (SC1) __hip_atomic_store(p, v[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // global_store_dword sc1
            else *p = v[r];
        }
    }
}

