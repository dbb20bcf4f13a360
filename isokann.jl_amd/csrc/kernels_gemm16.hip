// bf16-MFMA GEMMs of the wide-network (layered) TRAINING path: C5 of BASELINE.json names "wide MLP [512,512,512], bf16 MFMA
// path" -- bf16 operands, f32 accumulate, f32 master weights (engine option train_bf16).
//
// One kernel shape serves every product of an optimizer step.  The matrix cores want BOTH operands contiguous along the
// contraction index (v_mfma_f32_16x16x32_bf16: lane (row|col = l & 15, kgroup = l >> 4) holds 8 consecutive k), so every product
// is written as an "NT" GEMM   C[i][j] = sum_k A[i][k] * B[j][k]   over row-major images A [M][lda], B [Nn][ldb]:
//   forward   z_l[b][n]       = sum_k a_{l-1}[b][k] * Wt_l[n][k]          Wt = bf16 TRANSPOSED weight image, refreshed per step
//   dX        r_{l-1}[b][k]   = sum_n delta_l[b][n] * W_l(mem)[k][n]      Flux's column-major (out, in) IS [k][n]: no image needed
//   dW + db   G_l[k][n]       = sum_b aT_{l-1}[k][b] * deltaT_l[n][b]     transposed bf16 images written by the producers'
//                                                                         epilogues; aT carries a row of ones, so row K of G is
//                                                                         db -- and Flux keeps b_l right behind W_l: ONE store
// The products of a 1000-row batch are small (<= 2 GFLOP, operands L2-resident): what a step costs is the number of dependent
// launches and their latency, not MFMA time.  So: no LDS staging -- fragments go global/L2 -> VGPR in MFMA shape, one step
// ahead -- a 32 x 64 output tile per 256-thread workgroup whose four waves SPLIT K (256..504 workgroups for every product of
// C5: the whole chip, 8..16 k-steps per wave), one LDS reduction, and fused epilogues (bias + activation + the bf16 / transposed
// images the next products need), so that a layer is one launch forward and two backward.
//
// Float32 arithmetic (Gemm16Args.f32math; the default of the layered training path, option train_bf16 = 0): the same kernels with
// f32 fragments -- a lane's eight consecutive contraction elements are eight registers -- and eight v_mfma_f32_16x16x4_f32 per
// 32-deep step (step j contracts the elements 8 (lane / 16) + j: any bijection of the contraction index serves as long as both
// operands use it), f32 images instead of bf16 ones.  Exact Float32 products and sums, i.e. what a library sgemm computes here
// until round 3 -- with the fused epilogues and grouped launches of the bf16 path instead of a library call per product plus
// elementwise kernels around it.
#include <type_traits>

#include "isk_internal.h"
#include "opt_rule.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ float g_sigmoid(float z) {
    float e = __expf(-fabsf(z));
    float r = 1.0f / (1.0f + e);
    return z >= 0.f ? r : e * r;
}
__device__ __forceinline__ float g_act(int act, float z) {
    switch (act) {
        case ISK_ACT_SIGMOID: return g_sigmoid(z);
        case ISK_ACT_TANH: return tanhf(z);
        case ISK_ACT_RELU: return fmaxf(z, 0.f);
        case ISK_ACT_SWISH: return z * g_sigmoid(z);
        default: return z;
    }
}
__device__ __forceinline__ float g_dact(int act, float z) {
    switch (act) {
        case ISK_ACT_SIGMOID: { float a = g_sigmoid(z); return a * (1.f - a); }
        case ISK_ACT_TANH: { float a = tanhf(z); return 1.f - a * a; }
        case ISK_ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case ISK_ACT_SWISH: { float s = g_sigmoid(z); return s + z * s * (1.f - s); }
        default: return 1.f;
    }
}
__device__ __forceinline__ u16 to_bf16(float v) {
    const __bf16 b = (__bf16)v;                          // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    return __builtin_bit_cast(u16, b);
}

__device__ __forceinline__ void st_img(void* img, long long idx, float v, int f32) {
    if (f32) ((float*)img)[idx] = v; else ((u16*)img)[idx] = to_bf16(v);
}
// the Float32 fragment: eight consecutive contraction elements of one operand row (f32 in memory)
template <bool SLOW>
__device__ __forceinline__ f32x8 load_frag32(const void* base, long long off, int k0, int kmax) {
    if (SLOW) {
        f32x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (k0 + j < kmax) ? ((const float*)base)[off + j] : 0.f;
        return r;
    }
    const f32x4* p = (const f32x4*)((const float*)base + off);
    const f32x4 lo = p[0], hi = p[1];
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// 8 consecutive contraction elements of one operand row as an MFMA fragment; kmax: elements at k >= kmax read as 0 (only the
// unaligned / ragged slow path passes kmax < the padded length)
template <bool SLOW>
__device__ __forceinline__ bf16x8 load_frag(const void* base, const int F32, long long off, int k0, int kmax) {
    if (SLOW) {
        bf16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = 0.f;
            if (k0 + j < kmax) v = F32 ? ((const float*)base)[off + j] : (float)((const __bf16*)base)[off + j];
            r[j] = (__bf16)v;
        }
        return r;
    }
    if (F32) {
        const f32x4* p = (const f32x4*)((const float*)base + off);
        const f32x4 lo = p[0], hi = p[1];
        const f32x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_convertvector(v, bf16x8);
    }
    return *(const bf16x8*)((const __bf16*)base + off);
}

#define G16_TM 32
#define G16_TN 64
#define G16_LDR 65           // LDS row stride of the reduction tile (floats): the transposed pass reads down a column

// one 32 x 64 output tile (bx, by) of problem `a`; the operand element types are wave-uniform run-time flags (a grouped launch
// mixes products with bf16 and f32 operands)
// NW waves split the contraction (4 or 8: the k-loop of a wave is a chain of memory round trips -- a workgroup reads its operand
// rows exactly once, mostly from beyond L2, ~1 us per step with two steps in flight -- so the deep products take eight waves)
// BW (bf16 arithmetic, fast path): the B operand is an f32 array (dX reads the master weights); an f32 A operand takes the slow path
template <bool SLOW, bool F32 = false, int NW = 4, bool BW = false>
__device__ __forceinline__ void gemm16_tile(const Gemm16Args& a, const int bx, const int by, float (*red)[G16_TM][G16_LDR]) {
    using frag_t = typename std::conditional<F32, f32x8, bf16x8>::type;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 15, lq = lane >> 4;
    const int i0 = bx * G16_TM, j0 = by * G16_TN;
    const int AF32 = a.a_f32, BF32 = a.b_f32;
    long long offA[2], offB[4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) { int r = i0 + rt * 16 + li; r = r < a.M ? r : a.M - 1; offA[rt] = (long long)r * a.lda + lq * 8; }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) { int r = j0 + ct * 16 + li; r = r < a.Nn ? r : a.Nn - 1; offB[ct] = (long long)r * a.ldb + lq * 8; }
    // fast path: buffer loads -- (operand base in SGPRs) + (one 32-bit byte offset per fragment row) + (the step as a scalar
    // offset).  With 64-bit addresses per load the kernel needed ~250 registers per wave; the fragment sets themselves are 104
    // (Float32 arithmetic: 128), and at <= 128 (168) registers two of these workgroups share a CU with a side-stream featurizer wave
    constexpr int ea = F32 ? 4 : 2, eb = (F32 || BW) ? 4 : 2;                    // bytes per operand element
    __amdgpu_buffer_rsrc_t ra, rb;
    int voA[2], voB[4];
    if constexpr (!SLOW) {
        // descriptors based at the tile's first operand row (64-bit scalar arithmetic), lane offsets relative to it: an operand may
        // be larger than the 2^31 bytes a 32-bit offset reaches (full-batch training of a wide network on 10^6 rows).  A tile
        // that lies wholly in the padding rows (i0 >= M: it only writes zeros into the images) gets negative offsets, i.e. huge
        // unsigned ones: out of range, the loads return 0
        ra = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.A + (long long)i0 * a.lda * ea), 0, 0x7ffffffc, 0x00020000);
        rb = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.B + (long long)j0 * a.ldb * eb), 0, 0x7ffffffc, 0x00020000);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) voA[rt] = (int)(offA[rt] - (long long)i0 * a.lda) * ea;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) voB[ct] = (int)(offB[ct] - (long long)j0 * a.ldb) * eb;
    }
    auto bload = [&](__amdgpu_buffer_rsrc_t r, int vo, int so) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0));
    };
    // one operand fragment of step ks: 8 consecutive contraction elements (wide: an f32 operand, two 16-byte loads)
    auto bfrag = [&](__amdgpu_buffer_rsrc_t r, int vo, int ks, int esz, auto WIDE) -> frag_t {
        const int so = __builtin_amdgcn_readfirstlane(ks * 32 * esz);
        if constexpr (F32) {
            const f32x4 lo = bload(r, vo, so), hi = bload(r, vo + 16, so);
            return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
            if constexpr (decltype(WIDE)::value) {
                const f32x4 lo = bload(r, vo, so), hi = bload(r, vo + 16, so);
                const f32x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                return __builtin_convertvector(v, bf16x8);
            }
            return __builtin_bit_cast(bf16x8, bload(r, vo, so));
        }
    };
    f32x4 acc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = a.Kc / 32;
    // this wave's k-steps: wave, wave + NW, ...  Fragments are requested ONE step ahead (two register sets).  Round 3 kept three
    // sets (two steps ahead) at ~250 registers per wave; two sets fit 128, and what hides the round trips -- ~1 us each beside the
    // side-stream featurizer -- is then the other workgroups on the CU (two 4-wave or one 8-wave workgroup beside a featurizer wave,
    // four / two without it): C5-shard iteration 3.08 -> 2.90 ms (bf16), and the deep products split K over eight waves
    constexpr int NSET = F32 ? 1 : 2;
    frag_t fa[NSET][2], fb[NSET][4];
    auto loadset = [&](int ks, auto SLOT) {
        constexpr int sl = decltype(SLOT)::value;
        if (ks < nk) {
            if constexpr (!SLOW) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) fa[sl][rt] = bfrag(ra, voA[rt], ks, ea, std::false_type{});
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) fb[sl][ct] = bfrag(rb, voB[ct], ks, eb, std::integral_constant<bool, BW>{});
            } else if constexpr (F32) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) fa[sl][rt] = load_frag32<SLOW>(a.A, offA[rt] + ks * 32, ks * 32 + lq * 8, a.kvalid);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) fb[sl][ct] = load_frag32<SLOW>(a.B, offB[ct] + ks * 32, ks * 32 + lq * 8, a.kvalid);
            } else {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) fa[sl][rt] = load_frag<SLOW>(a.A, AF32, offA[rt] + ks * 32, ks * 32 + lq * 8, a.kvalid);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) fb[sl][ct] = load_frag<SLOW>(a.B, BF32, offB[ct] + ks * 32, ks * 32 + lq * 8, a.kvalid);
            }
        }
    };
    auto mma = [&](auto SLOT) {
        constexpr int sl = decltype(SLOT)::value;
        if constexpr (F32) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl][rt][j], fb[sl][ct][j], acc[rt][ct], 0, 0, 0);
        } else {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[sl][rt], fb[sl][ct], acc[rt][ct], 0, 0, 0);
        }
    };
    using S0 = std::integral_constant<int, 0>;
    int ks = wave;
    if constexpr (NSET == 1) {
        for (; ks < nk; ks += NW) { loadset(ks, S0{}); mma(S0{}); }
    } else {
        using S1 = std::integral_constant<int, NSET - 1>;
        loadset(ks, S0{});
        for (; ks < nk; ks += 2 * NW) {
            loadset(ks + NW, S1{});
            mma(S0{});
            if (ks + NW >= nk) break;
            loadset(ks + 2 * NW, S0{});
            mma(S1{});
        }
    }
    // C layout of the 16x16 tile: column = lane & 15, row = 4 * (lane >> 4) + r
    // (eight waves: two stages, so that the buffer stays at four tile sets -- waves 4..7 park their tiles, waves 0..3 add their own
    // to the same elements, which only they touch)
    if (NW == 4 || wave >= 4) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave & 3][rt * 16 + 4 * lq + r][ct * 16 + li] = acc[rt][ct][r];
    }
    __syncthreads();
    if constexpr (NW == 8) {
        if (wave < 4) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[wave][rt * 16 + 4 * lq + r][ct * 16 + li] += acc[rt][ct][r];
        }
        __syncthreads();
    }
    // pass 1: consecutive threads along j -- the K-split sum, the epilogue, the row-major stores.  Every global operand of the
    // epilogue (bias / previous pre-activations) is requested UNCONDITIONALLY, from clamped addresses, before the first use: behind
    // a per-element `if` the eight loads of a thread became eight dependent memory round trips (~1 us each: the whole kernel took
    // 12 us for a product that needs 2)
    const bool want_t = a.ot16 != nullptr;
    constexpr int NT = NW * 64, NP = G16_TM * G16_TN / NT;        // threads, elements per thread
    float ext[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int idx = tid + NT * p, row = idx >> 6, col = idx & 63;
        const int ic = (i0 + row) < a.M ? (i0 + row) : a.M - 1, jc = (j0 + col) < a.Nn ? (j0 + col) : a.Nn - 1;
        ext[p] = 0.f;
        if (a.epi == ISK_G16_FWD) { if (a.bias) ext[p] = a.bias[jc]; }
        else if (a.epi == ISK_G16_DX) ext[p] = a.zprev[(long long)ic * a.ldzp + jc];
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int idx = tid + NT * p, row = idx >> 6, col = idx & 63;
        const int i = i0 + row, j = j0 + col;
        float v = (red[0][row][col] + red[1][row][col]) + (red[2][row][col] + red[3][row][col]);
        const bool in = i < a.M && j < a.Nn;
        float o = 0.f;                                   // what the bf16 images receive (0 in the padding)
        if (a.epi == ISK_G16_FWD) {
            v += ext[p];
            const float av = g_act(a.act, v);
            if (in) {
                a.z[(long long)i * a.ldz + j] = v;
                o = av;
                if (a.o32) a.o32[(long long)i * a.ld32 + j] = o;
            }
        } else if (a.epi == ISK_G16_DX) {
            const float dv = v * g_dact(a.act, ext[p]);
            if (in) o = dv;
        } else {                                         // ISK_G16_DW: rows 0..K-1 are dW (Flux layout [k][n]), row K is db
            if (in) a.gout[(long long)i * a.gld + j] = v;
        }
        if (a.o16 && i < a.M16 && j < a.ld16) st_img(a.o16, (long long)i * a.ld16 + j, o, F32);
        if (want_t) red[0][row][col] = o;                // own element only: no other thread reads it before the barrier
    }
    if (!want_t) return;
    __syncthreads();
    // pass 2: consecutive threads along i -- the transposed bf16 image [j][i] (rows of it beyond Nn are never read)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int idx = tid + NT * p, col = idx >> 5, row = idx & 31;
        const int i = i0 + row, j = j0 + col;
        if (j < a.Nn && i < a.ldt) st_img(a.ot16, (long long)j * a.ldt + i, red[0][row][col], F32);
    }
}

// Register cap: left to itself hipcc spends ~250 registers per wave here (copies parked in AGPRs), which lets ONE 4-wave workgroup
// share a CU with the side-stream featurizer (its waves hold ~248 of a SIMD's 512) and no 8-wave workgroup at all.  128 (bf16) /
// 168 (Float32 fragments) registers are what the fragment sets need.
#define G16_WAVES_PER_EU(F32_) 4
template <bool SLOW, bool F32, int NW, bool BW>
__global__ __launch_bounds__(NW * 64, G16_WAVES_PER_EU(F32)) void gemm16_kernel(Gemm16Args a) {
    __shared__ float red[4][G16_TM][G16_LDR];
    ISK_LATENCY_CHAIN_PRIO();
    gemm16_tile<SLOW, F32, NW, BW>(a, blockIdx.x, blockIdx.y, red);
}

// several INDEPENDENT products in one launch (dW_l beside dX_l: both need only delta_l): a step of the layered training path is a
// chain of dependent launches, and two products that do not depend on each other should not wait for each other
template <bool F32, int NW>
__global__ __launch_bounds__(NW * 64, G16_WAVES_PER_EU(F32)) void gemm16_group_kernel(Gemm16Group g) {
    __shared__ float red[4][G16_TM][G16_LDR];
    ISK_LATENCY_CHAIN_PRIO();
    if (g.head_part) {
        // the head kernel's per-workgroup partials (dW_L | db_L | loss) summed by this launch's first blocks: sixteen lanes per
        // element, each takes every sixteenth workgroup's partial (independent loads, all in flight), then a fixed shuffle tree --
        // one thread per element walking the G partials one after the other took 26 us
        const long long e = ((long long)blockIdx.x * (NW * 64) + threadIdx.x) >> 4;
        const int l16 = threadIdx.x & 15;
        if (e <= g.head_len) {                                       // (uniform per 16-lane group)
            float s = 0.f;
            for (int w = l16; w < g.head_G; w += 16) s += g.head_part[(long long)w * g.head_stride + e];
            s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
            if (l16 == 0) {
                if (e < g.head_len) g.head_out[e] = s;
                else { g.head_loss[0] = s; g.head_loss[1] = g.head_rank_err ? (float)*g.head_rank_err : 0.f; }
            }
        }
    }
    int p = 0, b = blockIdx.x;
    while (p + 1 < g.n && b >= g.first[p + 1]) ++p;
    b -= g.first[p];
    // (a member with a ragged shape -- the last layer's one-column products -- takes the elementwise operand path: the choice is
    // uniform per workgroup; Float32 arithmetic is a kernel of its own: its fragments need twice the registers)
    if (g.slow[p]) gemm16_tile<true, F32, NW>(g.g[p], b % g.gx[p], b / g.gx[p], red);
    else if (!F32 && g.g[p].b_f32) gemm16_tile<false, F32, NW, !F32>(g.g[p], b % g.gx[p], b / g.gx[p], red);
    else gemm16_tile<false, F32, NW>(g.g[p], b % g.gx[p], b / g.gx[p], red);
}

static bool g16_grid(const Gemm16Args& a, int* gx, int* gy, bool* slow) {
    if (a.M < 1 || a.Nn < 1 || a.Kc < 32 || a.Kc % 32) return false;
    const int rows = a.M16 > a.M ? a.M16 : a.M;
    *gx = (rows + G16_TM - 1) / G16_TM; *gy = (a.Nn + G16_TN - 1) / G16_TN;
    // vector loads need 16-byte aligned rows; anything else (odd widths, a contraction that is not a multiple of 32) takes the
    // elementwise path
    if (a.f32math && !(a.a_f32 && a.b_f32)) return false;               // Float32 arithmetic reads f32 operands
    const int ea = a.a_f32 ? 4 : 8, eb = a.b_f32 ? 4 : 8;
    const bool a_al = ((uintptr_t)a.A % 16 == 0) && (a.lda % ea == 0), b_al = ((uintptr_t)a.B % 16 == 0) && (a.ldb % eb == 0);
    *slow = !a_al || !b_al || a.kvalid < a.Kc || (!a.f32math && a.a_f32);       // (bf16 arithmetic on an f32 A operand: no fast path)
    return true;
}

// waves per workgroup: eight where a wave would otherwise walk more than four 32-deep steps (option-free: both variants give the
// same sums up to the order of the K-split partials)
// (Float32 arithmetic: a fragment is eight registers, so ONE fragment set -- a 32-deep step is 64 MFMAs of 32 cycles, ~0.85 us,
// which a second wave on the SIMD fills with its own round trip: 72 registers, 3.75 -> 3.67 ms per C5-shard iteration against two sets
// at ~190 registers and one workgroup per CU.  Walking the contraction in 16-deep steps instead (four-register fragments) is slower,
// 236..250 against 209 us per optimizer step: twice the round trips.  bf16 with one set: 145..148 against 143 us)
static int g16_waves(const Gemm16Args& a) { return a.waves == 4 || a.waves == 8 ? a.waves : (a.Kc / 32 >= 16 ? 8 : 4); }

int isk_launch_gemm16(const Gemm16Args& a, hipStream_t st) {
    int gx, gy; bool slow;
    if (!g16_grid(a, &gx, &gy, &slow)) return ISK_EINVAL;
    void* args[] = {(void*)&a};
    const int nw = g16_waves(a);
    const bool bw = !a.f32math && a.b_f32 && !slow;
    const void* fn;
#define G16_PICK(NW_)                                                                                                              \
    (a.f32math ? (slow ? (const void*)gemm16_kernel<true, true, NW_, false> : (const void*)gemm16_kernel<false, true, NW_, false>)      \
               : (slow ? (const void*)gemm16_kernel<true, false, NW_, false>                                                           \
                       : (bw ? (const void*)gemm16_kernel<false, false, NW_, true> : (const void*)gemm16_kernel<false, false, NW_, false>)))
    if (nw == 8) fn = G16_PICK(8); else fn = G16_PICK(4);
#undef G16_PICK
    return hipLaunchKernel(fn, dim3(gx, gy), dim3(nw * 64), args, 0, st) == hipSuccess ? 0 : ISK_EHIP;
}

int isk_launch_gemm16_group(const Gemm16Args* list, int n, hipStream_t st, const Gemm16Head* head) {
    if (n < 1 || n > ISK_G16_MAXGROUP) return ISK_EINVAL;
    if (head && n < 2) return ISK_EINVAL;                              // (the side job rides on the grouped kernel)
    Gemm16Group g;
    g.n = n;
    g.head_part = nullptr;
    if (head) {
        g.head_part = head->part; g.head_stride = head->stride; g.head_G = head->G; g.head_len = head->len; g.head_out = head->out;
        g.head_loss = head->loss; g.head_rank_err = head->rank_err;
    }
    int total = 0, nw = 4;
    for (int p = 0; p < n; ++p) {
        int gx, gy; bool slow;
        if (!g16_grid(list[p], &gx, &gy, &slow)) return ISK_EINVAL;
        g.g[p] = list[p]; g.gx[p] = gx; g.first[p] = total; g.slow[p] = slow ? 1 : 0; g.f32[p] = list[p].f32math;
        if (list[p].f32math != list[0].f32math) return ISK_EINVAL;     // one arithmetic per launch
        total += gx * gy;
        if (g16_waves(list[p]) == 8) nw = 8;
    }
    if (n == 1) return isk_launch_gemm16(list[0], st);
    if (head && (long long)total * (nw * 64) <= 16 * (head->len + 1)) return ISK_EINVAL;   // (cannot happen: a product has far more threads than the last layer has weights)
    void* args[] = {(void*)&g};
    const void* fn = list[0].f32math ? (nw == 8 ? (const void*)gemm16_group_kernel<true, 8> : (const void*)gemm16_group_kernel<true, 4>)
                                     : (nw == 8 ? (const void*)gemm16_group_kernel<false, 8> : (const void*)gemm16_group_kernel<false, 4>);
    return hipLaunchKernel(fn, dim3(total), dim3(nw * 64), args, 0, st) == hipSuccess ? 0 : ISK_EHIP;
}

// ---------------------------------------------------------------------------------------------------------------------
// images around the GEMMs
// ---------------------------------------------------------------------------------------------------------------------
// batch rows (gathered by perm) -> xb [Bp][Kp] bf16 (A operand of the first layer's forward) and xt [K + 1 (+pad)][Bp] bf16 (its
// transposed image with the row of ones behind the K features: the first layer's dW + db operand).  One 32 x 32 tile per
// workgroup through LDS; rows >= B and columns >= K are written as zeros (they are contraction padding).
struct GatherArgs {
    const float* X; int ldx; const int* perm; int B, Bp, K, Kp; void* xb; void* xt; long long ldt;
    int gx;                                                            // tiles along the batch rows (linear block index = bx + gx * by)
    int f32;                                                           // the images are float arrays (Float32 arithmetic)
};
__device__ __forceinline__ void gather_tile(const GatherArgs& g, const int bx, const int by, float (*tile)[33]) {
    const float* __restrict__ X = g.X;
    const int* __restrict__ perm = g.perm;
    const int B = g.B, Bp = g.Bp, K = g.K, Kp = g.Kp, ldx = g.ldx;
    void* __restrict__ xb = g.xb;
    void* __restrict__ xt = g.xt;
    const long long ldt = g.ldt;
    const int f32 = g.f32;
    const int b0 = bx * 32, k0 = by * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 8 row groups
    // the four rows of a thread: indices, then values, requested together from clamped addresses (no load behind a branch: each
    // would be a memory round trip of its own)
    long long src[4];
    float val[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int b = b0 + ty + 8 * q;
        const int bc = b < B ? b : B - 1;
        src[q] = perm ? perm[bc] : bc;
    }
    const int kc = (k0 + tx) < K ? (k0 + tx) : K - 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) val[q] = X[src[q] * ldx + kc];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = ty + 8 * q;
        const int b = b0 + r, k = k0 + tx;
        float v = 0.f;
        if (b < B && k < K) v = val[q];
        else if (b < B && k == K) v = 1.f;                            // the ones row of the transposed image
        tile[r][tx] = v;
        if (b < Bp && k < Kp) st_img(xb, (long long)b * Kp + k, k < K ? v : 0.f, f32);
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, b = b0 + tx;
        if (k <= K && b < Bp) st_img(xt, (long long)k * ldt + b, tile[tx][r], f32);
    }
}
__global__ __launch_bounds__(256) void gather_images_kernel(GatherArgs g) {
    __shared__ float tile[32][33];
    ISK_LATENCY_CHAIN_PRIO();
    gather_tile(g, blockIdx.x % g.gx, blockIdx.x / g.gx, tile);
}
static GatherArgs gather_args(const float* X, int ldx, const int* perm, int B, int Bp, int K, int Kp, void* xb, void* xt, long long ldt, int f32,
                              int* nblocks) {
    GatherArgs g = {X, ldx, perm, B, Bp, K, Kp, xb, xt, ldt, (Bp + 31) / 32, f32};
    *nblocks = g.gx * ((K + 1 + 31) / 32);                             // covers k = K (the ones row)
    return g;
}
#ifdef ISK_EXPERIMENTAL
// the batch images of a whole EPOCH in one launch (blockIdx.y = batch: rows perm[y * B .. + B), images y strides further): the
// gather depends on the shuffle order only, so it need not sit in front of every optimizer step's chain
__global__ __launch_bounds__(256) void gather_epoch_kernel(GatherArgs g, long long xb_stride, long long xt_stride) {
    __shared__ float tile[32][33];
    const int y = blockIdx.y;
    g.perm += (long long)y * g.B;
    g.xb = (char*)g.xb + y * xb_stride;
    g.xt = (char*)g.xt + y * xt_stride;
    gather_tile(g, blockIdx.x % g.gx, blockIdx.x / g.gx, tile);
}
void isk_launch_gather_epoch(const float* X, int ldx, const int* perm, int B, int Bp, int K, int Kp, void* xb, void* xt, long long ldt,
                             int img_f32, int nbatch, long long xb_stride_bytes, long long xt_stride_bytes, hipStream_t st) {
    int nb;
    const GatherArgs g = gather_args(X, ldx, perm, B, Bp, K, Kp, xb, xt, ldt, img_f32, &nb);
    hipLaunchKernelGGL(gather_epoch_kernel, dim3(nb, nbatch), dim3(256), 0, st, g, xb_stride_bytes, xt_stride_bytes);
}
#endif
void isk_launch_gather_images(const float* X, int ldx, const int* perm, int B, int Bp, int K, int Kp, void* xb, void* xt, long long ldt,
                              int img_f32, hipStream_t st) {
    int nb;
    const GatherArgs g = gather_args(X, ldx, perm, B, Bp, K, Kp, xb, xt, ldt, img_f32, &nb);
    hipLaunchKernelGGL(gather_images_kernel, dim3(nb), dim3(256), 0, st, g);
}

// Wt_l[n][k] = bf16(W_l[n][k] * scale[k]) for every layer in one launch (forward B operands; and / or the same numbers in the
// fragment-linear order Wf_l the fused wide forward streams), W in Flux layout mem [k][n];
// scale = the LayerNorm scale for layer 0 (or nullptr).  32 x 32 tiles through LDS; columns k >= K are zero padding.
__device__ __forceinline__ void wt_tile(const WtImagesArgs& a, int blk, float (*tile)[33]) {
    int l = 0;
    while (l + 1 < a.nlayers && blk >= a.first_block[l + 1]) ++l;
    blk -= a.first_block[l];
    const int K = a.K[l], N = a.N[l], Kp = a.Kp[l];
    const int tk = (Kp + 31) / 32;
    const int k0 = (blk % tk) * 32, n0 = (blk / tk) * 32;
    const float* W = a.theta + a.off_W[l];
    const float* sc = (l == 0) ? a.scale0 : nullptr;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    float wv[4], sv[4];
    const int nc = (n0 + tx) < N ? (n0 + tx) : N - 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {                                     // unconditional, clamped: four loads in flight, not four round trips
        const int kc = (k0 + ty + 8 * q) < K ? (k0 + ty + 8 * q) : K - 1;
        wv[q] = W[(long long)kc * N + nc];
        sv[q] = sc ? sc[kc] : 1.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = ty + 8 * q;
        const int k = k0 + r, n = n0 + tx;
        tile[r][tx] = (k < K && n < N) ? wv[q] * sv[q] : 0.f;
    }
    __syncthreads();
    void* out = a.Wt[l];
    u16* frag = (u16*)a.Wf[l];
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, k = k0 + tx;
        if (n < N && k < Kp) {
            const u16 h = to_bf16(tile[tx][r]);
            if (out) { if (a.wt_f32) ((float*)out)[(long long)n * Kp + k] = tile[tx][r]; else ((u16*)out)[(long long)n * Kp + k] = h; }
            // fragment-linear image of the fused wide forward (kernels_fwd_wide.hip): 16-column block n / 16 of 32, step k / 32; lane
            // (n % 16) + 16 ((k % 32) / 8) holds features k % 8 of its 16 bytes
            if (frag) frag[((((long long)(k >> 5) * 32 + (n >> 4)) * 64 + ((n & 15) + 16 * ((k & 31) >> 3))) << 3) + (k & 7)] = h;
        }
    }
}
__global__ __launch_bounds__(256) void wt_images_kernel(WtImagesArgs a) {
    __shared__ float tile[32][33];
    ISK_LATENCY_CHAIN_PRIO();
    wt_tile(a, blockIdx.x, tile);
}
// the two image jobs in front of an optimizer step's first product -- the batch rows (independent of the parameters) and the weight
// images (behind the previous step's update) -- as ONE launch: a step is a chain of dependent launches, each worth ~5 us
__global__ __launch_bounds__(256) void prep_images_kernel(GatherArgs g, WtImagesArgs w, int ngather) {
    __shared__ float tile[32][33];
    ISK_LATENCY_CHAIN_PRIO();
    const int b = blockIdx.x;
    if (b < ngather) gather_tile(g, b % g.gx, b / g.gx, tile);
    else wt_tile(w, b - ngather, tile);
}
// The boundary between two optimizer steps of the wide-network path as ONE launch: the optimizer step of step s (every weight
// matrix tile by tile, the biases elementwise), the row-major weight images of step s + 1 written from the freshly updated tile
// (the same numbers wt_images_kernel would read back), and the batch images of step s + 1 (they depend on the shuffle order only).
// It replaces opt_step + prep_images: one dependent launch less per step.  Semantics of opt_step_kernel: a non-finite batch loss, a
// flagged rank or an earlier collapse of the epoch skips the update (the images then keep their -- still valid -- contents).
// No leading LayerNorm (its scale is folded into layer 0's image and is itself a parameter of the step): the caller keeps the two
// launches there.
__device__ __forceinline__ void opt_wt_tile(const OptArgs& oa, const WtImagesArgs& a, int blk, const bool bad, const float bc1, const float bc2,
                                            float (*tile)[33]) {
    int l = 0;
    while (l + 1 < a.nlayers && blk >= a.first_block[l + 1]) ++l;
    blk -= a.first_block[l];
    const int K = a.K[l], N = a.N[l], Kp = a.Kp[l];
    const int tk = (Kp + 31) / 32;
    const int k0 = (blk % tk) * 32, n0 = (blk / tk) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const bool adam = oa.kind == ISK_OPT_ADAM;
    long long idx[4];
    float th[4], gs[4], mi[4], vi[4];
    const int nc = (n0 + tx) < N ? (n0 + tx) : N - 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {                                     // every operand requested together, from clamped addresses
        const int kc = (k0 + ty + 8 * q) < K ? (k0 + ty + 8 * q) : K - 1;
        idx[q] = a.off_W[l] + (long long)kc * N + nc;
        th[q] = oa.theta[idx[q]]; gs[q] = oa.grad[idx[q]]; mi[q] = oa.m[idx[q]]; vi[q] = adam ? oa.v[idx[q]] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = ty + 8 * q;
        const int k = k0 + r, n = n0 + tx;
        const bool in = k < K && n < N;
        if (in && !bad) {
            opt_rule(oa, bc1, bc2, gs[q], th[q], mi[q], vi[q]);
            oa.theta[idx[q]] = th[q]; oa.m[idx[q]] = mi[q];
            if (adam) oa.v[idx[q]] = vi[q];
        }
        tile[r][tx] = in ? th[q] : 0.f;
    }
    if (bad) return;                                                  // (uniform) nothing changed: the images are current
    __syncthreads();
    void* out = a.Wt[l];
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, k = k0 + tx;
        if (n < N && k < Kp && out) {
            if (a.wt_f32) ((float*)out)[(long long)n * Kp + k] = tile[tx][r]; else ((u16*)out)[(long long)n * Kp + k] = to_bf16(tile[tx][r]);
        }
    }
}
struct OptPrepBias { int nlayers; long long off_b[ISK_MAX_LAYERS]; int n[ISK_MAX_LAYERS]; };
__global__ __launch_bounds__(256) void opt_prep_kernel(OptArgs oa, WtImagesArgs w, OptPrepBias bb, GatherArgs g, int nwt, int nbias, int ngather) {
    __shared__ float tile[32][33];
    ISK_LATENCY_CHAIN_PRIO();
    const int b = blockIdx.x;
    if (b >= nwt + nbias) { if (ngather > 0) gather_tile(g, (b - nwt - nbias) % g.gx, (b - nwt - nbias) / g.gx, tile); return; }
    const float bc1 = oa.bt ? (float)(1.0 - oa.bt[0]) : oa.bc1, bc2 = oa.bt ? (float)(1.0 - oa.bt[1]) : oa.bc2;
    const float lstep = oa.grad[oa.n], peer = oa.grad[oa.n + 1];
    const double collapsed = oa.loss_acc[1];
    const bool bad = (collapsed != 0.0) || !isfinite(lstep) || peer != 0.f;
    if (b < nwt) {
        opt_wt_tile(oa, w, b, bad, bc1, bc2, tile);
    } else {
        // the biases: element e of their concatenation
        int e = (b - nwt) * 256 + threadIdx.x, l = 0;
        while (l < bb.nlayers && e >= bb.n[l]) { e -= bb.n[l]; ++l; }
        if (l < bb.nlayers && !bad) {
            const long long i = bb.off_b[l] + e;
            float th = oa.theta[i], mi = oa.m[i], vi = oa.kind == ISK_OPT_ADAM ? oa.v[i] : 0.f;
            opt_rule(oa, bc1, bc2, oa.grad[i], th, mi, vi);
            oa.theta[i] = th; oa.m[i] = mi;
            if (oa.kind == ISK_OPT_ADAM) oa.v[i] = vi;
        }
    }
    if (b == 0 && threadIdx.x == 0) {                                 // the epoch's bookkeeping, as in opt_step_kernel
        double* const loss_acc = oa.loss_acc;
        double acc0 = loss_acc[0];
        if (peer != 0.f) loss_acc[3] = (double)peer;
        if (bad) loss_acc[1] = 1.0;
        else { acc0 += (double)lstep; loss_acc[0] = acc0; }
        if (oa.host_pub) {
            oa.host_pub[0] = acc0;
            oa.host_pub[1] = bad ? 1.0 : 0.0;
            ((int*)(oa.host_pub + 2))[0] = ((const int*)(loss_acc + 2))[0];
            oa.host_pub[3] = peer != 0.f ? (double)peer : loss_acc[3];
            __threadfence_system();
        }
    }
}
// next_perm == nullptr: no batch follows (the epoch's last step)
void isk_launch_opt_prep(const OptArgs& oa, WtImagesArgs& w, const long long* off_b, const float* X, int ldx, const int* next_perm, int B, int Bp,
                         int K, int Kp, void* xb, void* xt, long long ldt, int img_f32, hipStream_t st) {
    int nwt = 0;
    for (int l = 0; l < w.nlayers; ++l) { w.first_block[l] = nwt; nwt += ((w.Kp[l] + 31) / 32) * ((w.N[l] + 31) / 32); }
    OptPrepBias bb = {};
    bb.nlayers = w.nlayers;
    int nbe = 0;
    for (int l = 0; l < w.nlayers; ++l) { bb.off_b[l] = off_b[l]; bb.n[l] = w.N[l]; nbe += w.N[l]; }
    const int nbias = (nbe + 255) / 256;
    int ng = 0;
    GatherArgs g = {};
    if (next_perm) g = gather_args(X, ldx, next_perm, B, Bp, K, Kp, xb, xt, ldt, img_f32, &ng);
    hipLaunchKernelGGL(opt_prep_kernel, dim3(nwt + nbias + ng), dim3(256), 0, st, oa, w, bb, g, nwt, nbias, ng);
}

void isk_launch_wt_images(WtImagesArgs& a, hipStream_t st) {
    int total = 0;
    for (int l = 0; l < a.nlayers; ++l) { a.first_block[l] = total; total += ((a.Kp[l] + 31) / 32) * ((a.N[l] + 31) / 32); }
    hipLaunchKernelGGL(wt_images_kernel, dim3(total), dim3(256), 0, st, a);
}
void isk_launch_prep_images(const float* X, int ldx, const int* perm, int B, int Bp, int K, int Kp, void* xb, void* xt, long long ldt,
                            int img_f32, WtImagesArgs& w, hipStream_t st) {
    int nb;
    const GatherArgs g = gather_args(X, ldx, perm, B, Bp, K, Kp, xb, xt, ldt, img_f32, &nb);
    int total = 0;
    for (int l = 0; l < w.nlayers; ++l) { w.first_block[l] = total; total += ((w.Kp[l] + 31) / 32) * ((w.N[l] + 31) / 32); }
    hipLaunchKernelGGL(prep_images_kernel, dim3(nb + total), dim3(256), 0, st, g, w, nb);
}

// ---------------------------------------------------------------------------------------------------------------------
// The head of the backward pass as ONE launch: last layer forward (nout <= 4 dot products per row), loss, delta_L, delta_{L-1}
// (row-major and transposed images) and the per-workgroup partial sums of dW_L / db_L / the loss.  It replaces three launches of
// ~0 flops each (the ragged last-layer product on the elementwise operand path, loss_delta16, the grouped dW_L / dX_L pair: 19 us
// alone, 34 beside the featurizer at C5) -- round 3's first "head" kernel lost to them because a wave walked its eight rows one
// dependent round trip after the other (and a second attempt with 32 rows per workgroup lost too: 32 workgroups of lone waves,
// ~2500 instructions each); here a workgroup takes EIGHT rows (two per wave), every operand requested before its first use.  Everything is Float32 on the f32 master weights and the stored activations: at least as accurate as the bf16 products it
// replaces.  The partial sums are added up, in workgroup order, by the first blocks of the NEXT launch (Gemm16Group.head_*): no
// atomics, no seam, deterministic.
struct HeadArgs {
    const void* a; int lda, a_f32;                 // activations of the last hidden layer: image [Bp][lda] (bf16 | f32)
    const float* zprev; int ldzp, act;            // its pre-activations [B][ldzp] and activation
    const float* W; const float* bias; int K, N;  // last layer: Flux layout W[o + k N], bias [N]
    const float* target; const float* lossw; const int* perm; int B; float inv_div; int lastact;
    void* d16; int ld16; void* dt16; long long ldt; int img_f32;      // delta_{L-1}: [Bp][ld16] and transposed [K][ldt]
    float* part; long long pstride;               // [gridDim.x][pstride]: dW_L (K N, Flux order) | db_L (N) | loss (1)
};
#define HEAD_RW 2                                  /* rows per wave */
#define HEAD_ROWS (4 * HEAD_RW)                    /* rows per workgroup: many small workgroups -- a lone wave costs its instruction count */
static_assert(HEAD_ROWS == ISK_HEAD_ROWS, "isk_internal.h: ISK_HEAD_ROWS");
#define HEAD_MAXN 4
#define HEAD_MAXK 1024
__global__ __launch_bounds__(256) void head_kernel(HeadArgs h) {
    __shared__ float sW[HEAD_MAXK * HEAD_MAXN];
    __shared__ __attribute__((aligned(16))) float sBuf[4 * 512 * HEAD_MAXN];   // per-wave dW partials of a 512-wide chunk, then the chunk's delta tile [512][HEAD_ROWS]
    __shared__ float sS[4][HEAD_MAXN + 1];                           // per-wave db and loss
    ISK_LATENCY_CHAIN_PRIO();
    constexpr int RW = HEAD_RW;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int K = h.K, N = h.N;
    const int w0 = blockIdx.x * HEAD_ROWS;                           // the workgroup's first row
    const int i0 = w0 + RW * wave;                                   // this wave's first row
    const int nchunk = (K + 511) >> 9;
    for (int i = tid; i < K * N; i += 256) sW[i] = h.W[i];
    auto load_a = [&](int j, int c, float (&v)[8]) {
        const int r = i0 + j;
        const int k = (c << 9) + 8 * lane;
        const long long off = (long long)r * h.lda + k;                // rows < Bp exist in the image (zeros beyond B)
        const bool in = k < h.lda;
        if (h.a_f32) {
            const f32x4* p = (const f32x4*)((const float*)h.a + (in ? off : 0));
            const f32x4 lo = p[0], hi = p[1];
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] = in ? lo[q] : 0.f; v[4 + q] = in ? hi[q] : 0.f; }
        } else {
            const bf16x8 x = *(const bf16x8*)((const __bf16*)h.a + (in ? off : 0));
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = in ? (float)x[q] : 0.f;
        }
    };
    // the rows' sample indices and targets: requested with the first operands
    long long prow[RW];
    bool rv[RW];
#pragma unroll
    for (int j = 0; j < RW; ++j) {
        const int r = i0 + j;
        rv[j] = r < h.B;
        prow[j] = h.perm ? h.perm[rv[j] ? r : h.B - 1] : (rv[j] ? r : h.B - 1);
    }
    // ---- phase 1: z_L[r][o] = sum_k a[r][k] W[k][o]
    float acc[RW][HEAD_MAXN];
#pragma unroll
    for (int j = 0; j < RW; ++j)
#pragma unroll
        for (int o = 0; o < HEAD_MAXN; ++o) acc[j][o] = 0.f;
    __syncthreads();                                                 // sW
    for (int c = 0; c < nchunk; ++c) {
        float av[RW][8];
#pragma unroll
        for (int j = 0; j < RW; ++j) load_a(j, c, av[j]);
        const int k0 = (c << 9) + 8 * lane;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = k0 + q;
#pragma unroll
            for (int o = 0; o < HEAD_MAXN; ++o) {
                const float w = (o < N && k < K) ? sW[k * N + o] : 0.f;
#pragma unroll
                for (int j = 0; j < RW; ++j) acc[j][o] = fmaf(av[j][q], w, acc[j][o]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < RW; ++j)
#pragma unroll
        for (int o = 0; o < HEAD_MAXN; ++o) {
            float v = acc[j][o];
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) v += __shfl_xor(v, sh, 64);
            acc[j][o] = v;                                           // every lane: the row's dot product
        }
    // ---- phase 2: loss and delta_L (uniform per row)
    float dL[RW][HEAD_MAXN];
    float lsum = 0.f, dbs[HEAD_MAXN];
#pragma unroll
    for (int o = 0; o < HEAD_MAXN; ++o) dbs[o] = 0.f;
#pragma unroll
    for (int j = 0; j < RW; ++j) {
#pragma unroll
        for (int o = 0; o < HEAD_MAXN; ++o) {
            float d = 0.f;
            if (o < N && rv[j]) {
                const float zz = acc[j][o] + h.bias[o];
                const float w = h.lossw[o];
                const float res = (g_act(h.lastact, zz) - h.target[prow[j] * N + o]) * w;
                lsum += res * res;
                d = 2.f * res * w * h.inv_div * g_dact(h.lastact, zz);
            }
            dL[j][o] = d;
            dbs[o] += d;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int o = 0; o < HEAD_MAXN; ++o) sS[wave][o] = dbs[o];
        sS[wave][HEAD_MAXN] = lsum;
    }
    // ---- phase 3: per chunk -- dW_L partials (this wave's rows), delta_{L-1} images
    float* const part = h.part + (long long)blockIdx.x * h.pstride;
    float* const sG = sBuf + wave * (512 * HEAD_MAXN);
    for (int c = 0; c < nchunk; ++c) {
        const int k0 = (c << 9) + 8 * lane;
        float av[RW][8];
        f32x4 zp[RW][2];
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            load_a(j, c, av[j]);
            const int r = i0 + j;
            const int rc = r < h.B ? r : h.B - 1;
            const int kc = k0 + 8 <= h.ldzp ? k0 : 0;                  // (columns beyond the padded width: a dummy, masked below)
            const f32x4* p = (const f32x4*)(h.zprev + (long long)rc * h.ldzp + kc);
            zp[j][0] = p[0]; zp[j][1] = p[1];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int o = 0; o < HEAD_MAXN; ++o) {
                float g = 0.f;
#pragma unroll
                for (int j = 0; j < RW; ++j) g = fmaf(av[j][q], dL[j][o], g);
                sG[(8 * lane + q) * HEAD_MAXN + o] = g;
            }
        // delta_{L-1}[r][k] = (sum_o delta_L[r][o] W[k][o]) act'(z_{L-1}[r][k])
        float dx[RW][8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = k0 + q;
            float wq[HEAD_MAXN];
#pragma unroll
            for (int o = 0; o < HEAD_MAXN; ++o) wq[o] = (o < N && k < K) ? sW[k * N + o] : 0.f;
#pragma unroll
            for (int j = 0; j < RW; ++j) {
                float sacc = 0.f;
#pragma unroll
                for (int o = 0; o < HEAD_MAXN; ++o) sacc = fmaf(dL[j][o], wq[o], sacc);
                dx[j][q] = k < K ? sacc * g_dact(h.act, zp[j][q >> 2][q & 3]) : 0.f;
            }
        }
        if (k0 < h.ld16) {                                           // row-major image: 8 contiguous k per lane and row
#pragma unroll
            for (int j = 0; j < RW; ++j) {
                const long long o16 = (long long)(i0 + j) * h.ld16 + k0;
                if (h.img_f32) {
                    f32x4* p = (f32x4*)((float*)h.d16 + o16);
                    p[0] = (f32x4){dx[j][0], dx[j][1], dx[j][2], dx[j][3]}; p[1] = (f32x4){dx[j][4], dx[j][5], dx[j][6], dx[j][7]};
                } else {
                    bf16x8 v;
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = (__bf16)dx[j][q];
                    *(bf16x8*)((__bf16*)h.d16 + o16) = v;
                }
            }
        }
        __syncthreads();                                             // the four waves' sG
        for (int i = tid; i < 512 * N; i += 256) {                   // element (kk, o) of the chunk, Flux order (k N + o)
            const int kk = i / N, o = i - kk * N, k = (c << 9) + kk;
            const int e = kk * HEAD_MAXN + o;
            if (k < K) part[(long long)k * N + o] = (sBuf[e] + sBuf[512 * HEAD_MAXN + e]) + (sBuf[2 * 512 * HEAD_MAXN + e] + sBuf[3 * 512 * HEAD_MAXN + e]);
        }
        __syncthreads();                                             // sBuf becomes the delta tile [512][HEAD_ROWS]
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int j = 0; j < RW; ++j) sBuf[(8 * lane + q) * HEAD_ROWS + RW * wave + j] = dx[j][q];
        __syncthreads();
        for (int kk = tid; kk < 512; kk += 256) {                    // transposed image: HEAD_ROWS = 8 contiguous rows per k
            const int k = (c << 9) + kk;
            if (k < K) {
                const f32x4 lo = *(const f32x4*)(sBuf + kk * HEAD_ROWS), hi = *(const f32x4*)(sBuf + kk * HEAD_ROWS + 4);
                const long long ot = (long long)k * h.ldt + w0;
                if (h.img_f32) { f32x4* p = (f32x4*)((float*)h.dt16 + ot); p[0] = lo; p[1] = hi; }
                else {
                    bf16x8 v;
#pragma unroll
                    for (int t = 0; t < 4; ++t) { v[t] = (__bf16)lo[t]; v[4 + t] = (__bf16)hi[t]; }
                    *(bf16x8*)((__bf16*)h.dt16 + ot) = v;
                }
            }
        }
        __syncthreads();                                             // before the next chunk overwrites sBuf
    }
    if (tid <= N) {                                                  // db_L (N) and the loss partial
        const int o = tid < N ? tid : HEAD_MAXN;
        part[(long long)K * N + tid] = (sS[0][o] + sS[1][o]) + (sS[2][o] + sS[3][o]);
    }
}
bool isk_head_ok(int K, int N, int B, int Bp, int lda, int ldzp) {
    return N >= 1 && N <= HEAD_MAXN && K >= 1 && K <= HEAD_MAXK && lda % 8 == 0 && lda >= K && ldzp % 8 == 0 && ldzp >= K && Bp % HEAD_ROWS == 0 && B >= 1 &&
           Bp / HEAD_ROWS <= ISK_HEAD_MAXG;
}
void isk_launch_head(const void* a, int lda, int a_f32, const float* zprev, int ldzp, int act, const float* W, const float* bias, int K, int N,
                     const float* target, const float* lossw, const int* perm, int B, int Bp, float inv_div, int lastact, void* d16, int ld16,
                     void* dt16, long long ldt, int img_f32, float* part, long long pstride, hipStream_t st) {
    HeadArgs h = {a, lda, a_f32, zprev, ldzp, act, W, bias, K, N, target, lossw, perm, B, inv_div, lastact, d16, ld16, dt16, ldt, img_f32, part, pstride};
    hipLaunchKernelGGL(head_kernel, dim3(Bp / HEAD_ROWS), dim3(256), 0, st, h);   // (Bp is a multiple of 32)
}

// delta_L and the batch loss from the last layer's pre-activations (as loss_delta_kernel of kernels_layered.hip), plus the
// bf16 images of delta_L: d16 [Bp][ld16] (A operand of dX) and dt16 [nout][Bp] (B operand of dW + db)
__global__ __launch_bounds__(1024) void loss_delta16_kernel(const float* __restrict__ z, int ld, const float* __restrict__ target,
                                                            const float* __restrict__ lossw, const int* __restrict__ perm, int B, int Bp,
                                                            int nout, float inv_div, int lastact, void* __restrict__ d16, int ld16,
                                                            void* __restrict__ dt16, long long ldt, int img_f32, float* __restrict__ loss_slot,
                                                            const int* __restrict__ rank_err) {
    __shared__ float red[16];
    float ls = 0.f;
    // rows n < Bp of the nout live columns only: the padding columns of d16 are never read (the dX product masks k >= nout), its
    // rows >= B neither (operand rows are clamped); the columns [B, Bp) of dt16 ARE contraction padding of dW and must be zeros
    const long long tot = (long long)Bp * nout;
    for (long long e = threadIdx.x; e < tot; e += 1024) {
        const int n = (int)(e / nout), o = (int)(e % nout);
        float d = 0.f;
        if (n < B) {
            const long long prow = perm ? perm[n] : n;
            const float zz = z[(long long)n * ld + o];
            const float w = lossw[o];
            const float r = (g_act(lastact, zz) - target[prow * nout + o]) * w;
            ls += r * r;
            d = 2.f * r * w * inv_div * g_dact(lastact, zz);
        }
        st_img(d16, (long long)n * ld16 + o, d, img_f32);
        st_img(dt16, (long long)o * ldt + n, d, img_f32);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ls += __shfl_xor(ls, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ls;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += red[i];
        loss_slot[0] = t;
        loss_slot[1] = rank_err ? (float)*rank_err : 0.f;
    }
}
void isk_launch_loss_delta16(const float* z, int ld, const float* target, const float* lossw, const int* perm, int B, int Bp, int nout,
                             float inv_div, int lastact, void* d16, int ld16, void* dt16, long long ldt, int img_f32, float* loss_slot,
                             const int* rank_err, hipStream_t st) {
    hipLaunchKernelGGL(loss_delta16_kernel, dim3(1), dim3(1024), 0, st, z, ld, target, lossw, perm, B, Bp, nout, inv_div, lastact,
                       d16, ld16, dt16, ldt, img_f32, loss_slot, rank_err);
}

