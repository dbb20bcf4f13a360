// Fused bf16 forward for WIDE networks (hidden widths <= 512: C5's densenet [2000,512,512,512,1]): the Koopman forward of
// forward_mode 2 (bf16 operands, f32 accumulate -- the arithmetic BASELINE.json names for C5) as ONE kernel in which the
// activations of a 128-row block never leave the CU.  The layer-by-layer path it replaces wrote and re-read a [rows][512]
// activation buffer per layer through HBM (8.4 GB per launch at C5-shard size against 3.3 GB of features) and re-streamed the
// sample rows per column block.
//
// Workgroup = 8 waves = 128 rows; wave (rg = w >> 2, cg = w & 3) owns rows [64 rg, +64) x columns {256 h + 64 cg + 16 c + li : h < 2,
// c < 4} of a layer's (padded) 512 outputs: 4 x 8 accumulator tiles of v_mfma_f32_16x16x32_bf16 = 128 registers.
//   layer 1   K-loop over the features, 32 per step.  BOTH tiles of a step go global -> LDS directly (global_load_lds_dwordx4, no
//             staging registers): the W tile (512 columns x 32, bf16 image [n][k], 32 KiB) and the A tile -- 128 rows x 32
//             features, either bf16 as the featurizer wrote them for this path (8 KiB; FeatArgs.out_bf16) or f32 (16 KiB, converted
//             when the fragments are read).  16-byte chunks are XOR-swizzled through the DMA's SOURCE addresses so that every
//             fragment read is conflict-free.  Three ring slots, tiles requested TWO steps ahead, ONE raw s_barrier per step
//             behind a counted s_waitcnt vmcnt(N) (N = the DMA instructions of one tile: the newest tile stays in flight).
//   layer l>1 A = the previous layer's activations, resident in LDS as bf16 [128][512] (chunks XOR-swizzled by row); the outputs
//             stay in registers until every wave has finished reading, then overwrite the activations in place.  The W tiles of
//             the two column halves are streamed one after the other through a two-slot ring of 16 KiB tiles (the activations
//             take 128 of the 160 KiB), one tile ahead.
//   last      nout <= 16 outputs per row: plain f32 dot products against the f32 weights, four lanes per row.
// What bounds it is how many bytes a CU takes in per MFMA: the accumulators of a 128 x 512 block fill half the register file, so
// the W tiles are re-streamed from L2 per 128 rows (320 B per row and step with bf16 rows); see DESIGN.md section 3 for the
// measured ablations.
// LDS: layer 1: [W ring 3 x 32 KiB | A ring 3 x 8 (16) KiB] <= 144 KiB; afterwards [activations 128 KiB | W ring 2 x 16 KiB] = 160 KiB.
#include <type_traits>

#include "isk_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

#define WF_BM 128
#define WF_NP 512
#define WF_ACT_BYTES (WF_BM * WF_NP * 2)                 // 131072
#define WF_W1_SLOT (512 * 64)                            // layer-1 W tile: 512 columns x 32 bf16
#define WF_A1_SLOT_F32 (WF_BM * 128)                     // layer-1 A tile: 128 rows x 32 f32
#define WF_A1_SLOT_B16 (WF_BM * 64)                      //                 128 rows x 32 bf16
#define WF_W2_SLOT (256 * 64)                            // later layers: 256 columns x 32 bf16
#define WF_LDS_BYTES (WF_ACT_BYTES + 2 * WF_W2_SLOT)     // 163840 >= 3 * (WF_W1_SLOT + WF_A1_SLOT_F32) = 147456

__device__ __forceinline__ float wf_sigmoid(float z) {
    float e = __expf(-fabsf(z));
    float r = 1.0f / (1.0f + e);
    return z >= 0.f ? r : e * r;
}
// the hidden activations of this kernel feed bf16 operands (8 mantissa bits): the sigmoid is the plain 1 / (1 + 2^(-z log2 e)) on
// the hardware exp2 / rcp (four instructions; an overflowing 2^x gives 1 / inf = 0, the right limit), not the branch-free
// two-sided form of the f32 paths (twice the instructions for accuracy that the rounding to bf16 discards)
template <int ACT>
__device__ __forceinline__ float wf_act(float z) {
    if (ACT == ISK_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * z));
    if (ACT == ISK_ACT_TANH) return tanhf(z);
    if (ACT == ISK_ACT_RELU) return fmaxf(z, 0.f);
    if (ACT == ISK_ACT_SWISH) return z * wf_sigmoid(z);
    return z;
}
__device__ __forceinline__ float wf_act_rt(int act, float z) {
    switch (act) {
        case ISK_ACT_SIGMOID: return wf_sigmoid(z);
        case ISK_ACT_TANH: return tanhf(z);
        case ISK_ACT_RELU: return fmaxf(z, 0.f);
        case ISK_ACT_SWISH: return z * wf_sigmoid(z);
        default: return z;
    }
}
// byte offset of column `col` of row `row` inside the swizzled activation image
__device__ __forceinline__ int act_off(int row, int col) { return row * (WF_NP * 2) + ((((col >> 3) ^ (row & 15))) << 4) + ((col & 7) << 1); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// one wave-instruction: 64 lanes x 16 B from per-lane global addresses to the contiguous KiB at `dst` (lane L lands at dst + 16 L)
__device__ __forceinline__ void glds16(const void* src, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
}

#define WF_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
// diagnostic builds only (timing valid, results not): -DWF_ABLATE=bits -- 1: the A-tile loads of layer 1 re-read ONE tile (cache
// resident: no HBM stream), 2: likewise the layer-1 W-tile loads, 4: no MFMAs (the fragment reads stay), 8: no LDS-DMA at all
#ifndef WF_ABLATE
#define WF_ABLATE 0
#endif

template <int ACT, bool XB16>
__global__ __launch_bounds__(512) void wide_fwd_bf16_kernel(WideFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // wave-uniform: addresses built from it stay in SGPRs
    const int li = lane & 15, lq = lane >> 4;
    const int rg = wave >> 2, cg = wave & 3;
    const long long nblk = (a.S + WF_BM - 1) / WF_BM;
    // Register budget (two waves per SIMD: 256): uniform 64-bit bases + 32-bit lane offsets, LDS offsets as instruction immediates
    // on top of one base per stream -- per-tile address registers next to the accumulators made hipcc spill into the K-loops.
    // bf16 tiles are [row | column][4 x 16 B]; the 16-byte part index is XORed with f((index >> 2) & 3), f = {0, 2, 3, 1}, through the
    // DMA's SOURCE addresses: with that f the sixteen lanes the LDS serves together for a ds_read_b128 ({lq0: li 0-3, 12-15;
    // lq1: li 4-11} and its mirror images) fall on sixteen different 4-bank groups -- unswizzled, the fragment reads were 2-way conflicts
    auto wswz = [](unsigned x) { return ((x << 1) ^ ((x >> 1) * 3u)) & 3u; };
    const unsigned frag_part = (((unsigned)lq ^ wswz((unsigned)(li >> 2) & 3u)) << 4);     // this lane's 16 B inside a 64-byte row of a bf16 tile
    const unsigned w_ld_off = (unsigned)((64 * cg + li) * 64) + frag_part;                 // B fragment of column tile (h = 0, c = 0)
    const unsigned src_part = ((unsigned)(lane & 3) ^ wswz((unsigned)(lane >> 4) & 3u)) * 8u;     // features this lane's 16 B hold in a 16-row DMA piece
    const unsigned a16_ld_off = (unsigned)((64 * rg + li) * 64) + frag_part;               // bf16 rows: A fragment of row tile 0
    const unsigned a_swz = (unsigned)(li >> 1);                                            // f32 rows: ((row >> 1) & 7) of every row this lane reads
    const unsigned a32_ld_lo = (unsigned)((64 * rg + li) * 128) + (((2u * lq) ^ a_swz) << 4);      // features 8 lq .. 8 lq + 3 of row tile 0
    const unsigned a32_ld_hi = (unsigned)((64 * rg + li) * 128) + (((2u * lq + 1u) ^ a_swz) << 4); // features 8 lq + 4 .. + 7
    constexpr int A1_SLOT = XB16 ? WF_A1_SLOT_B16 : WF_A1_SLOT_F32;
    for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long long row0 = blk * WF_BM;
        f32x4 acc[4][8];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // ------------------------------------------------------------------ layer 1: A and W tiles by LDS-DMA ----------------
        {
            unsigned char* const Wr = lds;                           // [3][512 cols][64 B]
            unsigned char* const Ar = lds + 3 * WF_W1_SLOT;          // [3][128 rows][64 | 128 B]
            const int nk = a.Kp[0] / 32;
            const int Kp = a.Kp[0];
            // W: 16 columns x 64 B per wave-instruction (lane L -> column 16 g + L / 4, features src_part ..), groups g = 4 wave + i.
            const unsigned char* wb = (const unsigned char*)a.Wt[0] + (size_t)(64 * wave) * Kp * 2;     // uniform
            const unsigned w_g_off = (unsigned)(((lane >> 2) * Kp + src_part) * 2);
            const unsigned wstep = (unsigned)(16 * Kp * 2);
            // A, bf16 rows: ONE instruction per wave (rows 16 wave + L / 4, as a W piece).  A, f32 rows: two (8 rows x 128 B each:
            // lane L -> row 16 wave + 8 j + L / 8, PHYSICAL chunk L % 8 holding logical chunk (L % 8) ^ ((row >> 1) & 7)).
            // Rows past the end re-read the last row (never stored).
            const int xes = XB16 ? 2 : 4;
            const unsigned char* xb = (const unsigned char*)a.X + (size_t)row0 * a.Fp * xes;             // uniform
            unsigned a_g_off[2];
            if (XB16) {
                long long xr = row0 + 16 * wave + (lane >> 2);
                xr = xr < a.S ? xr : a.S - 1;
                a_g_off[0] = (unsigned)(((xr - row0) * a.Fp + src_part) * 2);
                a_g_off[1] = 0;
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = 16 * wave + 8 * j + (lane >> 3);
                    long long xr = row0 + row;
                    xr = xr < a.S ? xr : a.S - 1;
                    a_g_off[j] = (unsigned)(((xr - row0) * a.Fp + 4 * ((lane & 7) ^ ((row >> 1) & 7))) * 4);
                }
            }
            auto issue = [&](int ks, int slot) {
                if (WF_ABLATE & 8) return;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    glds16((WF_ABLATE & 2) ? wb + w_g_off : wb + (size_t)i * wstep + (size_t)ks * 64 + w_g_off, Wr + slot * WF_W1_SLOT + (4 * wave + i) * 1024);
                const size_t kx = (WF_ABLATE & 1) ? 0 : (size_t)ks * 32 * xes;
                if (XB16) {
                    glds16(xb + kx + a_g_off[0], Ar + slot * A1_SLOT + (16 * wave) * 64);
                } else {
                    glds16(xb + kx + a_g_off[0], Ar + slot * A1_SLOT + (16 * wave) * 128);
                    glds16(xb + kx + a_g_off[1], Ar + slot * A1_SLOT + (16 * wave + 8) * 128);
                }
            };
            issue(0, 0);
            if (nk > 1) issue(1, 1);
            int slot = 0;
            for (int ks = 0; ks < nk; ++ks) {
                // tile ks has landed once every wave's share has: this wave's DMA instructions of it are older than those of tile
                // ks + 1 (vmcnt retires in order), then the barrier.  The barrier also says every wave is through the MFMAs of
                // step ks - 1, i.e. through its reads of slot (ks + 2) % 3, which the next request overwrites
                if (ks + 1 < nk) { if (XB16) WF_WAIT_VM(5); else WF_WAIT_VM(6); } else WF_WAIT_VM(0);
                __builtin_amdgcn_s_barrier();
                // The two waves of a SIMD (w and w + 4) run the same program in lockstep after every barrier: both would sit in the
                // DMA-issue phase together (the matrix pipe idle), then contend for the matrix pipe together.  Waves 4..7 therefore
                // request the next tile in the MIDDLE of their MFMAs: while one wave of the SIMD issues its DMA, the other computes.
                const bool late = wave >= 4;
                if (!late && ks + 2 < nk) issue(ks + 2, slot == 0 ? 2 : slot - 1);
                const unsigned char* As = Ar + slot * A1_SLOT;
                const unsigned char* Ws = Wr + slot * WF_W1_SLOT + w_ld_off;
                bf16x8 fa[4];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    if (XB16) {
                        fa[rt] = *(const bf16x8*)(As + rt * 1024 + a16_ld_off);
                    } else {
                        const f32x4 lo = *(const f32x4*)(As + rt * 2048 + a32_ld_lo), hi = *(const f32x4*)(As + rt * 2048 + a32_ld_hi);
                        fa[rt] = __builtin_convertvector(__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7), bf16x8);
                    }
                }
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) {
                    if (ct == 4 && late && ks + 2 < nk) issue(ks + 2, slot == 0 ? 2 : slot - 1);
                    const bf16x8 fb = *(const bf16x8*)(Ws + (ct >> 2) * 16384 + (ct & 3) * 1024);
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) {
                        if (WF_ABLATE & 4) asm volatile("" :: "v"(fa[rt]), "v"(fb));
                        else acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[rt], fb, acc[rt][ct], 0, 0, 0);
                    }
                }
                slot = slot == 2 ? 0 : slot + 1;
            }
            __syncthreads();                                 // every wave is through its last reads: the rings are dead
        }
        // ------------------------------------------------------------------ hidden layers: epilogue, then the next K-loop from LDS
        // epilogue of layer l: the outputs overwrite the rings / the previous activations in place.  Element (rt, ct = 4 h + c, r) is
        // row 64 rg + 16 rt + 4 lq + r, column 256 h + 64 cg + 16 c + li; its 16-byte chunk index (column / 8) is XORed with
        // row & 15 = 4 lq + r.
        auto epilogue = [&](int l) {
            const float* bias = a.bias[l];
            const int Nl = a.N[l];
            float bv[8];
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {                 // the eight bias loads together, from clamped addresses (no load behind a branch)
                const int col = 256 * (ct >> 2) + 64 * cg + 16 * (ct & 3) + li;
                bv[ct] = bias[col < Nl ? col : Nl - 1];
            }
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
                const int col = 256 * (ct >> 2) + 64 * cg + 16 * (ct & 3) + li;
                const bool live = col < Nl;                  // padded columns feed zeros into the next contraction
                const unsigned cbase = (unsigned)((64 * rg + 4 * lq) * (WF_NP * 2) + ((col & 7) << 1));
                const unsigned chunk = (unsigned)(col >> 3);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = wf_act<ACT>(acc[rt][ct][r] + bv[ct]);
                        v = live ? v : 0.f;
                        const __bf16 h = (__bf16)v;
                        *(u16*)(lds + cbase + (rt * 16 + r) * (WF_NP * 2) + ((chunk ^ (unsigned)(4 * lq + r)) << 4)) = __builtin_bit_cast(u16, h);
                        acc[rt][ct][r] = 0.f;
                    }
            }
            __syncthreads();
        };
        epilogue(0);
        for (int l = 1; l < a.nhidden; ++l) {
            // K-loop of layer l: contraction over the (padded) width of layer l - 1; the W tiles of the two column halves follow each
            // other through ONE two-slot ring (step s: half s / nk, k-step s % nk), one tile ahead, across the half boundary
            unsigned char* const Wr = lds + WF_ACT_BYTES;    // [2][256 cols][64 B]
            const int nk = a.Kp[l] / 32;
            const int Kp = a.Kp[l];
            const int nsteps = 2 * nk;
            // W: two column groups of 16 per wave and step (g = 2 wave + i)
            const unsigned char* wb = (const unsigned char*)a.Wt[l] + (size_t)(32 * wave) * Kp * 2;        // uniform
            const unsigned w_g_off = (unsigned)(((lane >> 2) * Kp + src_part) * 2);
            const unsigned a_row_off = (unsigned)((64 * rg + li) * (WF_NP * 2));
            auto issue = [&](int s2, int slot) {
                if (WF_ABLATE & 8) return;
                const int h = s2 >= nk ? 1 : 0, ks = s2 - h * nk;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    glds16(wb + (size_t)(256 * h + 16 * i) * Kp * 2 + (size_t)ks * 64 + w_g_off, Wr + slot * WF_W2_SLOT + (2 * wave + i) * 1024);
            };
            issue(0, 0);
            int slot = 0, s2 = 0;
            auto half = [&](auto H) {
                constexpr int h = decltype(H)::value;
                for (int ks = 0; ks < nk; ++ks, ++s2) {
                    WF_WAIT_VM(0);                            // this wave's share of tile s2 (requested a whole step ago) has landed ...
                    __builtin_amdgcn_s_barrier();             // ... and everybody's; everybody is through its reads of the other slot
                    const bool late = wave >= 4;              // as in layer 1: the SIMD partners request their tiles at different points
                    if (!late && s2 + 1 < nsteps) issue(s2 + 1, slot ^ 1);
                    const unsigned ach = a_row_off + ((((unsigned)(ks * 4 + lq)) ^ (unsigned)li) << 4);       // row & 15 == li for every row tile
                    const unsigned char* Ws = Wr + slot * WF_W2_SLOT + w_ld_off;
                    bf16x8 fa[4];
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) fa[rt] = *(const bf16x8*)(lds + ach + rt * (16 * WF_NP * 2));
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (c == 2 && late && s2 + 1 < nsteps) issue(s2 + 1, slot ^ 1);
                        const bf16x8 fb = *(const bf16x8*)(Ws + c * 1024);
#pragma unroll
                        for (int rt = 0; rt < 4; ++rt) {
                            if (WF_ABLATE & 4) asm volatile("" :: "v"(fa[rt]), "v"(fb));
                            else acc[rt][4 * h + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[rt], fb, acc[rt][4 * h + c], 0, 0, 0);
                        }
                    }
                    slot ^= 1;
                }
            };
            half(std::integral_constant<int, 0>{});
            half(std::integral_constant<int, 1>{});
            __syncthreads();                                 // every wave is through its reads of the activations
            epilogue(l);
        }
        // ------------------------------------------------------------------ last layer: nout <= 16 f32 dot products per row
        {
            // four threads per row, each takes every fourth 16-byte chunk (8 hidden units) of the row: one ds_read_b128 and two
            // 16-byte weight loads per 8 multiply-adds (element by element this phase took as long as a whole hidden layer's epilogue)
            const int row = tid >> 2, q = tid & 3;
            const int K = a.Klast;                            // = width of the last hidden layer (<= 512)
            const int nch = (K + 7) >> 3;
            const unsigned char* arow = lds + row * (WF_NP * 2);
            if (a.nout == 1 && (K & 7) == 0) {
                float s = 0.f;
#pragma unroll 1
                for (int c = q; c < nch; c += 4) {
                    const bf16x8 av = *(const bf16x8*)(arow + ((c ^ (row & 15)) << 4));
                    const f32x4 w0 = *(const f32x4*)(a.Wlast + 8 * c), w1 = *(const f32x4*)(a.Wlast + 8 * c + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { s = fmaf((float)av[j], w0[j], s); s = fmaf((float)av[4 + j], w1[j], s); }
                }
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                if (q == 0 && row0 + row < a.S) a.out[(row0 + row) * a.out_ld] = wf_act_rt(a.lastact, s + a.blast[0]);
            } else {
                for (int o = 0; o < a.nout; ++o) {
                    float s = 0.f;
#pragma unroll 1
                    for (int c = q; c < nch; c += 4) {
                        const bf16x8 av = *(const bf16x8*)(arow + ((c ^ (row & 15)) << 4));
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int k = 8 * c + j;
                            if (k < K) s = fmaf((float)av[j], a.Wlast[(long long)k * a.nout + o], s);   // Flux (out, in) column-major: (o, k) at o + k nout
                        }
                    }
                    s += __shfl_xor(s, 1, 64);
                    s += __shfl_xor(s, 2, 64);
                    if (q == 0 && row0 + row < a.S) a.out[(row0 + row) * a.out_ld + o] = wf_act_rt(a.lastact, s + a.blast[o]);
                }
            }
        }
        __syncthreads();                                      // the next block's rings overwrite the activations
    }
}

int isk_wide_fwd_ok(const WideFwdArgs& a) {
    if (a.nhidden < 1 || a.nhidden >= ISK_MAX_LAYERS || a.nout < 1 || a.nout > 16) return 0;
    for (int l = 0; l < a.nhidden; ++l)
        if (a.N[l] < 1 || a.N[l] > WF_NP || a.Kp[l] % 32 || (l > 0 && a.Kp[l] > WF_NP)) return 0;
    return a.Klast == a.N[a.nhidden - 1] && a.Fp == a.Kp[0];
}

int isk_launch_wide_fwd(const WideFwdArgs& a, hipStream_t st) {
    if (a.S <= 0) return 0;
    if (!isk_wide_fwd_ok(a)) return ISK_EUNSUPPORTED;
    const void* fn;
#define WFPICK(A_) (a.x_bf16 ? (const void*)wide_fwd_bf16_kernel<A_, true> : (const void*)wide_fwd_bf16_kernel<A_, false>)
    switch (a.act) {
        case ISK_ACT_SIGMOID: fn = WFPICK(ISK_ACT_SIGMOID); break;
        case ISK_ACT_TANH: fn = WFPICK(ISK_ACT_TANH); break;
        case ISK_ACT_RELU: fn = WFPICK(ISK_ACT_RELU); break;
        case ISK_ACT_SWISH: fn = WFPICK(ISK_ACT_SWISH); break;
        default: fn = WFPICK(ISK_ACT_IDENTITY); break;
    }
#undef WFPICK
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, WF_LDS_BYTES) != hipSuccess) return ISK_EHIP;
    const long long nblk = (a.S + WF_BM - 1) / WF_BM;
    const unsigned grid = (unsigned)(nblk < ISK_NUM_CU ? nblk : ISK_NUM_CU);
    void* args[] = {(void*)&a};
    return hipLaunchKernel(fn, dim3(grid), dim3(512), args, WF_LDS_BYTES, st) == hipSuccess ? 0 : ISK_EHIP;
}
