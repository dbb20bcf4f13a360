// Split-bf16 ("bf16x3") variant of the fused MLP forward for gfx950.
//
// gfx950's f32-input MFMA runs at 1/16 of the bf16 MFMA rate.  Layer 1 (99 % of the flops) is therefore
// computed as an error-compensated bf16 product: every f32 operand is split into three bf16 planes
//      a = a0 + a1 + a2,   a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)      (24 mantissa bits)
// and a.b is accumulated in f32 from the six largest cross terms
//      a0b0 + (a0b1 + a1b0) + (a0b2 + a2b0 + a1b1)          dropped: a1b2, a2b1, a2b2  (<= 2^-24 relative)
// i.e. 6 x v_mfma_f32_16x16x32_bf16 per 16x16x32 block = 16/6 = 2.67x the f32 MFMA throughput at
// f32-equivalent accuracy (measured on K = 1712 dot products: rms error 6.0e-7 vs 5.1e-7 for an f32 BLAS).
//
// Structure (differs from the f32 kernel because the matrix pipe is now 2.7x faster and the vector
// memory path would saturate): the three weight planes of a 32-feature chunk (3*NT KiB, fragment-linear,
// pre-split by fold_bf16_kernel) are staged through LDS once per workgroup and shared by its 8 waves
// (two per SIMD: while one converts / waits, the other feeds the matrix pipe)
// (register-staged T14 pipeline: global load one chunk ahead, ds_write after the compute, one barrier per
// chunk); the sample rows stay private to a wave: f32 global -> VGPR in fragment shape, three chunks
// ahead, split into planes in registers right before use.  Tail layers, bias and activation are the f32
// code shared with the f32 kernel (fwd_common.hpp).
#include <cstdlib>
#include "fwd_common.hpp"

typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Planes {
    bf16x8 p0, p1, p2;
};

#ifndef ISK_ABLATE
#define ISK_ABLATE 0
#endif
__device__ __forceinline__ Planes split3(const f32x8 v) {
    Planes s;
#if ISK_ABLATE == 1
    {   // ablation: no conversion arithmetic
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 b = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
        s.p0 = __builtin_bit_cast(bf16x8, b); s.p1 = s.p0; s.p2 = s.p0;
        asm volatile("" :: "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
        return s;
    }
#endif
    s.p0 = __builtin_convertvector(v, bf16x8);
    const f32x8 r1 = v - __builtin_convertvector(s.p0, f32x8);
    s.p1 = __builtin_convertvector(r1, bf16x8);
    const f32x8 r2 = r1 - __builtin_convertvector(s.p1, f32x8);
    s.p2 = __builtin_convertvector(r2, bf16x8);
    return s;
}

// fp32 fragment-linear layer-1 weights [chunk16][nt][64][4]  ->  bf16 planes [chunk32][3][nt][64][8]
// lane = (kq << 4) | o_local holds W[o][k = 32*c + 8*kq + j], j = 0..7: the B operand of v_mfma_f32_16x16x32_bf16
__global__ __launch_bounds__(256) void fold_bf16_kernel(const float* __restrict__ Wt, __bf16* __restrict__ out, int nchunk32,
                                                        int nt, int nchunk16) {
    const long long total = (long long)nchunk32 * nt * 64 * 8;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int j = e & 7;
        const int lane = (e >> 3) & 63;
        const long long ct = e >> 9;
        const int t = (int)(ct % nt);
        const int c32 = (int)(ct / nt);
        const int li = lane & 15, kq = lane >> 4;
        const int c16 = 2 * c32 + (kq >> 1);
        const int k16 = 8 * (kq & 1) + j;
        float w = 0.f;
        if (c16 < nchunk16) w = Wt[(((long long)c16 * nt + t) * 64 + (((k16 >> 2) << 4) | li)) * 4 + (k16 & 3)];
        const __bf16 b0 = (__bf16)w;
        const float r1 = w - (float)b0;
        const __bf16 b1 = (__bf16)r1;
        const __bf16 b2 = (__bf16)(r1 - (float)b1);
        const long long base = (long long)c32 * 3 * nt;
        out[((base + 0 * nt + t) * 64 + lane) * 8 + j] = b0;
        out[((base + 1 * nt + t) * 64 + lane) * 8 + j] = b1;
        out[((base + 2 * nt + t) * 64 + lane) * 8 + j] = b2;
    }
}

void isk_launch_fold_bf16(const float* Wt, void* out, int Kp, int nt, hipStream_t st) {
    const int nchunk32 = (Kp + 31) / 32, nchunk16 = Kp / 16;
    long long total = (long long)nchunk32 * nt * 64 * 8;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fold_bf16_kernel, dim3(blocks), dim3(256), 0, st, Wt, (__bf16*)out, nchunk32, nt, nchunk16);
}

template <int NT, bool TWO>
__device__ __forceinline__ void mma6(const Planes& x0, const Planes& x1, const bf16x8* __restrict__ wl, int lane,
                                     f32x4 (&acc0)[NT], f32x4 (&acc1)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const bf16x8 w0 = wl[(0 * NT + t) * 64 + lane];
        const bf16x8 w1 = wl[(1 * NT + t) * 64 + lane];
        const bf16x8 w2 = wl[(2 * NT + t) * 64 + lane];
        // smallest terms first
        acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0.p1, w1, acc0[t], 0, 0, 0);
        if (TWO) acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1.p1, w1, acc1[t], 0, 0, 0);
        acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0.p2, w0, acc0[t], 0, 0, 0);
        if (TWO) acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1.p2, w0, acc1[t], 0, 0, 0);
        acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0.p0, w2, acc0[t], 0, 0, 0);
        if (TWO) acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1.p0, w2, acc1[t], 0, 0, 0);
        acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0.p1, w0, acc0[t], 0, 0, 0);
        if (TWO) acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1.p1, w0, acc1[t], 0, 0, 0);
        acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0.p0, w1, acc0[t], 0, 0, 0);
        if (TWO) acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1.p0, w1, acc1[t], 0, 0, 0);
        acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0.p0, w0, acc0[t], 0, 0, 0);
        if (TWO) acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1.p0, w0, acc1[t], 0, 0, 0);
    }
}

// One pass = the whole feature axis for this wave's (up to) two 16-row units.  The weight planes of one
// 32-feature chunk (3*NT KiB) are staged per barrier, two LDS buffers.  Register-staged pipeline: the
// global loads of chunk c+2 are issued right after the barrier that ends chunk c and are written to
// LDS after the compute of chunk c+1.  The sample fragments come from HBM and are fetched XDEPTH chunks
// ahead (ablation: with a 2-chunk lead the kernel loses 29 % to exposed HBM latency).
#ifndef XDEPTH
#define XDEPTH 3
#endif

template <int NT, bool TWO, int NTHR>
__device__ __forceinline__ void layer1_pass(const MlpFwdArgs& a, const uint4* __restrict__ W16, uint4* wbuf0, uint4* wbuf1,
                                            const float* x0, const float* x1, int nch, int lane, f32x4 (&acc0)[NT],
                                            f32x4 (&acc1)[NT], bool active) {
    constexpr int STAGE16 = 3 * NT * 64;                 // 16-byte units staged per chunk (this block's tiles)
    constexpr int PER_T = (STAGE16 + NTHR - 1) / NTHR;
    const int tid = threadIdx.x;
    const int last = nch - 1;
    // global image: [chunk][plane][nt_total tiles][64]; the block stages tiles [tile0, tile0 + NT) of each plane
    const int ntt = a.dense_mode ? a.w_nt_total : NT;
    const long long gchunk = (long long)3 * ntt * 64;
    int goff[PER_T];
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
        const int idx = i * NTHR + tid;
        const int p = idx / (NT * 64), rem = idx - p * (NT * 64);
        goff[i] = (p * ntt + (a.dense_mode ? a.w_tile0 : 0)) * 64 + rem;
    }
    uint4 wreg[PER_T];
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
        const int idx = i * NTHR + tid;
        wreg[i] = (idx < STAGE16) ? W16[goff[i]] : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();                                     // the previous pass's epilogue scratch aliases the buffers
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
        const int idx = i * NTHR + tid;
        if (idx < STAGE16) wbuf0[idx] = wreg[i];
    }
    __syncthreads();
    if (nch > 1) {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int idx = i * NTHR + tid;
            wreg[i] = (idx < STAGE16) ? W16[gchunk + goff[i]] : make_uint4(0, 0, 0, 0);
        }
    }
    f32x8 xr0[XDEPTH + 1], xr1[XDEPTH + 1];
#pragma unroll
    for (int i = 0; i < XDEPTH; ++i) {
        const int ci = i < last ? i : last;
        xr0[i] = *(const f32x8*)(x0 + ci * 32);
        if (TWO) xr1[i] = *(const f32x8*)(x1 + ci * 32);
    }
    for (int c0 = 0; c0 < nch; c0 += XDEPTH + 1) {
#pragma unroll
        for (int s = 0; s <= XDEPTH; ++s) {
            const int c = c0 + s;
            if (c < nch) {                                            // workgroup-uniform
                {
                    const int cp = c + XDEPTH < last ? c + XDEPTH : last;
                    xr0[(s + XDEPTH) % (XDEPTH + 1)] = *(const f32x8*)(x0 + cp * 32);
                    if (TWO) xr1[(s + XDEPTH) % (XDEPTH + 1)] = *(const f32x8*)(x1 + cp * 32);
                }
                const uint4* cur = (c & 1) ? wbuf1 : wbuf0;
                uint4* nxt = (c & 1) ? wbuf0 : wbuf1;
                if (active) {
                    const Planes p0 = split3(xr0[s]);
                    Planes p1 = p0;
                    if (TWO) p1 = split3(xr1[s]);
                    mma6<NT, TWO>(p0, p1, (const bf16x8*)cur, lane, acc0, acc1);
                }
                if (c + 1 < nch) {
#pragma unroll
                    for (int i = 0; i < PER_T; ++i) {
                        const int idx = i * NTHR + tid;
                        if (idx < STAGE16) nxt[idx] = wreg[i];
                    }
                }
                __syncthreads();
                if (c + 2 < nch) {
#pragma unroll
                    for (int i = 0; i < PER_T; ++i) {
                        const int idx = i * NTHR + tid;
                        wreg[i] = (idx < STAGE16) ? W16[(long long)(c + 2) * gchunk + goff[i]] : make_uint4(0, 0, 0, 0);
                    }
                }
            }
        }
    }
}

template <int NT, int NWV>
__global__ __launch_bounds__(NWV * 64) void mlp_fwd_bf16x3_kernel(MlpFwdArgs a, const uint4* __restrict__ W16, long long units,
                                                                  long long units_per_wave) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE_FLOATS = 3 * NT * 64 * 4;                   // one chunk per buffer
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    uint4* wbuf0 = (uint4*)(smem + a.tailw_floats);
    uint4* wbuf1 = (uint4*)(smem + a.tailw_floats + STAGE_FLOATS);
    // the per-wave epilogue scratch ALIASES the weight buffers: it is only used between the last barrier
    // of a pass and the first barrier of the next one
    float* scrA = smem + a.tailw_floats + (size_t)wave * 16 * (a.stride_a + a.stride_b);
    float* scrB = scrA + 16 * a.stride_a;
    stage_tail_weights(a, smem);
    __syncthreads();
    const int nch = a.Fp / 32;
    const long long gw = (long long)blockIdx.x * NWV + wave;
    const long long ustart = gw * units_per_wave;
    long long uend = ustart + units_per_wave;
    if (uend > units) uend = units;
    const long long npass = (units_per_wave + 1) / 2;              // identical for every wave: barriers inside
    for (long long p = 0; p < npass; ++p) {
        const long long u = ustart + 2 * p;
        const bool active = u < uend;                              // wave-uniform
        const bool two = (u + 1 < uend);
        long long r0 = u * 16 + li, r1 = r0 + 16;
        if (r0 > a.S - 1) r0 = a.S - 1;
        if (r1 > a.S - 1) r1 = a.S - 1;
        if (r0 < 0) r0 = 0;
        if (a.rows) { r0 = a.rows[r0]; r1 = a.rows[r1]; }
        const float* x0 = a.X + r0 * (long long)a.Fp + lq * 8;
        const float* x1 = a.X + r1 * (long long)a.Fp + lq * 8;
        f32x4 acc0[NT], acc1[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { acc0[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        if (two) layer1_pass<NT, true, NWV * 64>(a, W16, wbuf0, wbuf1, x0, x1, nch, lane, acc0, acc1, active);
        else layer1_pass<NT, false, NWV * 64>(a, W16, wbuf0, wbuf1, x0, x1, nch, lane, acc0, acc1, active);
        if (active) unit_tail<NT>(a, acc0, u * 16, scrA, scrB, smem, lane);
        if (two) unit_tail<NT>(a, acc1, (u + 1) * 16, scrA, scrB, smem, lane);
    }
}

template <int NT, int NWV>
static size_t bf16_lds_bytes(const MlpFwdArgs& a) {
    const size_t wb = (size_t)2 * 3 * NT * 256, scr = (size_t)NWV * 16 * (a.stride_a + a.stride_b);
    return ((size_t)a.tailw_floats + (wb > scr ? wb : scr)) * sizeof(float);
}

template <int NT, int NWV>
static void launch_bf16_nw(const MlpFwdArgs& a, const void* W16, hipStream_t st) {
    const long long units = (a.S + 15) / 16;
    const long long waves_max = (long long)ISK_NUM_CU * NWV;
    long long upw = (units + waves_max - 1) / waves_max;
    if (upw < 1) upw = 1;
    const long long nwaves = (units + upw - 1) / upw;
    const long long blocks = (nwaves + NWV - 1) / NWV;
    const size_t lds = bf16_lds_bytes<NT, NWV>(a);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)mlp_fwd_bf16x3_kernel<NT, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((mlp_fwd_bf16x3_kernel<NT, NWV>), dim3((unsigned)blocks), dim3(NWV * 64), lds, st, a, (const uint4*)W16, units, upw);
}

template <int NT>
static int launch_bf16_nt(const MlpFwdArgs& a, const void* W16, hipStream_t st) {
    // two waves per SIMD by default: one converts / stages while the other feeds the matrix pipe;
    // fall back to one wave per SIMD when the per-wave LDS scratch would not fit
    static const int nwv = [] { const char* v = getenv("ISK_BF16_WAVES"); return v ? atoi(v) : 8; }();
    const size_t cap = 158 * 1024;
    if (nwv != 4 && bf16_lds_bytes<NT, 8>(a) <= cap) { launch_bf16_nw<NT, 8>(a, W16, st); return 0; }
    if (bf16_lds_bytes<NT, 4>(a) <= cap) { launch_bf16_nw<NT, 4>(a, W16, st); return 0; }
    return ISK_EUNSUPPORTED;
}

int isk_launch_mlp_fwd_bf16x3(const MlpFwdArgs& a, const void* W16, hipStream_t st) {
    if (a.S <= 0) return 0;
    if (a.Fp % 32 != 0) return ISK_EUNSUPPORTED;
    switch (a.nt1) {
        case 1: return launch_bf16_nt<1>(a, W16, st);
        case 2: return launch_bf16_nt<2>(a, W16, st);
        case 3: return launch_bf16_nt<3>(a, W16, st);
        case 4: return launch_bf16_nt<4>(a, W16, st);
        case 5: return launch_bf16_nt<5>(a, W16, st);
        case 6: return launch_bf16_nt<6>(a, W16, st);
        case 7: return launch_bf16_nt<7>(a, W16, st);
        case 8: return launch_bf16_nt<8>(a, W16, st);
        case 9: return launch_bf16_nt<9>(a, W16, st);
        case 10: return launch_bf16_nt<10>(a, W16, st);
        default: return ISK_EUNSUPPORTED;
    }
    return 0;
}
