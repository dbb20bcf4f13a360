// Fused bf16 forward for WIDE networks (hidden widths <= 512: C5's densenet [2000,512,512,512,1]): the Koopman forward of
// forward_mode 2 (bf16 operands, f32 accumulate -- the arithmetic BASELINE.json names for C5) as ONE kernel in which the
// activations of a 128-row block never leave the CU.  The layer-by-layer path it replaces wrote and re-read a [rows][512]
// activation buffer per layer through HBM (8.4 GB per launch at C5-shard size against 3.3 GB of features) and re-streamed the
// sample rows per column block.
//
// Workgroup = 8 waves = 96 rows; wave (rg = w >> 2, cg = w & 3) owns rows [48 rg, +48) x columns {256 h + 64 cg + 16 c + li : h < 2,
// c < 4} of a layer's (padded) 512 outputs: 3 x 8 accumulator tiles of v_mfma_f32_16x16x32_bf16 = 96 registers.
//   layer 1   K-loop over the features, 32 per step.  BOTH tiles of a step go global -> LDS directly (global_load_lds_dwordx4, no
//             staging registers): the W tile (512 columns x 32, bf16 image [n][k], 32 KiB) and the A tile (96 rows x 32 features,
//             12 KiB, kept as f32 and converted to bf16 when the fragments are read; 16-byte chunks XOR-swizzled through the SOURCE
//             addresses so that a fragment read is conflict-free).  Three ring slots, tiles requested TWO steps ahead, ONE raw
//             s_barrier per step behind a counted s_waitcnt vmcnt(N) (N = the LDS-DMA instructions of one tile: the newest tile
//             stays in flight across the barrier).  The first version -- two slots, __syncthreads(), i.e. vmcnt(0) in front of
//             every barrier -- exposed a full HBM round trip per step: 2.1 ms per C5-shard launch; before that, address registers
//             spilled into the loop (3.5 ms).
//   layer l>1 A = the previous layer's activations, resident in LDS as bf16 [96][512] (16-byte chunks XOR-swizzled by row);
//             the outputs stay in registers until every wave has finished reading, then overwrite the activations in place.  The
//             W tiles of the two column halves are streamed one after the other through a three-slot ring of 16 KiB tiles.
//   last      nout <= 16 outputs per row: plain f32 dot products against the f32 weights, four lanes per row.
// LDS: layer 1: [W ring 3 x 32 KiB | A ring 3 x 12 KiB] = 132 KiB; afterwards [activations 96 KiB | W ring 3 x 16 KiB] = 144 KiB.
#include <type_traits>

#include "isk_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

#define WF_BM 96
#define WF_NP 512
#define WF_ACT_BYTES (WF_BM * WF_NP * 2)                 // 98304
#define WF_W1_SLOT (512 * 64)                            // layer-1 W tile: 512 columns x 32 bf16
#define WF_A1_SLOT (WF_BM * 128)                         // layer-1 A tile: 96 rows x 32 f32
#define WF_W2_SLOT (256 * 64)                            // later layers: 256 columns x 32 bf16
#define WF_LDS_BYTES (WF_ACT_BYTES + 3 * WF_W2_SLOT)     // 147456 >= 3 * (WF_W1_SLOT + WF_A1_SLOT) = 135168

__device__ __forceinline__ float wf_sigmoid(float z) {
    float e = __expf(-fabsf(z));
    float r = 1.0f / (1.0f + e);
    return z >= 0.f ? r : e * r;
}
template <int ACT>
__device__ __forceinline__ float wf_act(float z) {
    if (ACT == ISK_ACT_SIGMOID) return wf_sigmoid(z);
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

template <int ACT>
__global__ __launch_bounds__(512) void wide_fwd_bf16_kernel(WideFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // wave-uniform: addresses built from it stay in SGPRs
    const int li = lane & 15, lq = lane >> 4;
    const int rg = wave >> 2, cg = wave & 3;
    const long long nblk = (a.S + WF_BM - 1) / WF_BM;
    // Register budget (two waves per SIMD: 256): uniform 64-bit bases + 32-bit lane offsets, LDS offsets as instruction immediates
    // on top of one base per stream -- per-tile address registers next to the accumulators made hipcc spill into the K-loops.
    // W tiles are [column][4 x 16 B]; the 16-byte part index is XORed with f((column >> 2) & 3), f = {0, 2, 3, 1}, through the DMA's
    // SOURCE addresses: with that f the sixteen lanes the LDS serves together for a ds_read_b128 ({lq0: li 0-3, 12-15; lq1: li 4-11}
    // and its mirror images) fall on sixteen different 4-bank groups -- unswizzled, the B-fragment reads were 2-way conflicts
    auto wswz = [](unsigned x) { return ((x << 1) ^ ((x >> 1) * 3u)) & 3u; };
    const unsigned w_ld_off = (unsigned)((64 * cg + li) * 64) + (((unsigned)lq ^ wswz((unsigned)(li >> 2) & 3u)) << 4);   // B fragment of column tile (h = 0, c = 0)
    const unsigned w_src_part = ((unsigned)(lane & 3) ^ wswz((unsigned)(lane >> 4) & 3u)) * 8u;     // features this lane's 16 B hold (column 16 g + lane / 4)
    const unsigned a_swz = (unsigned)(li >> 1);                                          // ((row >> 1) & 7) of every row this lane reads: rows 48 rg + 16 rt + li
    const unsigned a_ld_lo = (unsigned)((48 * rg + li) * 128) + (((2u * lq) ^ a_swz) << 4);       // features 8 lq .. 8 lq + 3 of row tile 0
    const unsigned a_ld_hi = (unsigned)((48 * rg + li) * 128) + (((2u * lq + 1u) ^ a_swz) << 4);  // features 8 lq + 4 .. + 7
    for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long long row0 = blk * WF_BM;
        f32x4 acc[3][8];
#pragma unroll
        for (int rt = 0; rt < 3; ++rt)
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // ------------------------------------------------------------------ layer 1: A and W tiles by LDS-DMA ----------------
        {
            unsigned char* const Wr = lds;                           // [3][512 cols][64 B]
            unsigned char* const Ar = lds + 3 * WF_W1_SLOT;          // [3][96 rows][128 B], chunks swizzled
            const int nk = a.Kp[0] / 32;
            const int Kp = a.Kp[0];
            // W: 16 columns x 64 B per wave-instruction (lane L -> column 16 g + L / 4, features 8 (L % 4) ..), groups g = 4 wave + i.
            const unsigned char* wb = (const unsigned char*)a.Wt[0] + (size_t)(64 * wave) * Kp * 2;     // uniform
            const unsigned w_g_off = (unsigned)(((lane >> 2) * Kp + w_src_part) * 2);
            const unsigned wstep = (unsigned)(16 * Kp * 2);
            // A: 8 rows x 128 B per wave-instruction (lane L -> row L / 8, PHYSICAL chunk L % 8 holding logical chunk
            // (L % 8) ^ ((row >> 1) & 7)); wave w stages rows 12 w .. 12 w + 11: one full instruction and one of 32 lanes
            const unsigned char* xb = (const unsigned char*)(a.X + row0 * a.Fp);                         // uniform
            unsigned a_g_off[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = 12 * wave + 8 * j + (lane >> 3);     // (lanes >= 32 of j = 1 are masked off at the instruction)
                long long xr = row0 + row;
                xr = xr < a.S ? xr : a.S - 1;
                a_g_off[j] = (unsigned)(((xr - row0) * a.Fp + 4 * ((lane & 7) ^ ((row >> 1) & 7))) * 4);
            }
            auto issue = [&](int ks, int slot) {
                if (WF_ABLATE & 8) return;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    glds16((WF_ABLATE & 2) ? wb + w_g_off : wb + (size_t)i * wstep + (size_t)ks * 64 + w_g_off, Wr + slot * WF_W1_SLOT + (4 * wave + i) * 1024);
                glds16((WF_ABLATE & 1) ? xb + a_g_off[0] : xb + (size_t)ks * 128 + a_g_off[0], Ar + slot * WF_A1_SLOT + (12 * wave) * 128);
                if (lane < 32) glds16((WF_ABLATE & 1) ? xb + a_g_off[1] : xb + (size_t)ks * 128 + a_g_off[1], Ar + slot * WF_A1_SLOT + (12 * wave + 8) * 128);
            };
            issue(0, 0);
            if (nk > 1) issue(1, 1);
            int slot = 0;
            for (int ks = 0; ks < nk; ++ks) {
                // tile ks has landed once every wave's share has: this wave's six DMA instructions of it are older than the six of
                // tile ks + 1 (vmcnt retires in order), then the barrier.  The barrier also says every wave is through the MFMAs of
                // step ks - 1, i.e. through its reads of slot (ks + 2) % 3, which the next request overwrites
                if (ks + 1 < nk) WF_WAIT_VM(6); else WF_WAIT_VM(0);
                __builtin_amdgcn_s_barrier();
                if (ks + 2 < nk) issue(ks + 2, slot == 0 ? 2 : slot - 1);
                const unsigned char* As = Ar + slot * WF_A1_SLOT;
                const unsigned char* Ws = Wr + slot * WF_W1_SLOT + w_ld_off;
                bf16x8 fa[3];
#pragma unroll
                for (int rt = 0; rt < 3; ++rt) {
                    const f32x4 lo = *(const f32x4*)(As + rt * 2048 + a_ld_lo), hi = *(const f32x4*)(As + rt * 2048 + a_ld_hi);
                    fa[rt] = __builtin_convertvector(__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7), bf16x8);
                }
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) {
                    const bf16x8 fb = *(const bf16x8*)(Ws + (ct >> 2) * 16384 + (ct & 3) * 1024);
#pragma unroll
                    for (int rt = 0; rt < 3; ++rt) {
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
        // row 48 rg + 16 rt + 4 lq + r, column 256 h + 64 cg + 16 c + li; its 16-byte chunk index (column / 8) is XORed with
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
                const unsigned cbase = (unsigned)((48 * rg + 4 * lq) * (WF_NP * 2) + ((col & 7) << 1));
                const unsigned chunk = (unsigned)(col >> 3);
#pragma unroll
                for (int rt = 0; rt < 3; ++rt)
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
            // other through ONE three-slot ring (step s: half s / nk, k-step s % nk), so the pipeline runs across the half boundary
            unsigned char* const Wr = lds + WF_ACT_BYTES;    // [3][256 cols][64 B]
            const int nk = a.Kp[l] / 32;
            const int Kp = a.Kp[l];
            const int nsteps = 2 * nk;
            // W: two column groups of 16 per wave and step (g = 2 wave + i)
            const unsigned char* wb = (const unsigned char*)a.Wt[l] + (size_t)(32 * wave) * Kp * 2;        // uniform
            const unsigned w_g_off = (unsigned)(((lane >> 2) * Kp + w_src_part) * 2);
            const unsigned a_row_off = (unsigned)((48 * rg + li) * (WF_NP * 2));
            auto issue = [&](int s2, int slot) {
                if (WF_ABLATE & 8) return;
                const int h = s2 >= nk ? 1 : 0, ks = s2 - h * nk;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    glds16(wb + (size_t)(256 * h + 16 * i) * Kp * 2 + (size_t)ks * 64 + w_g_off, Wr + slot * WF_W2_SLOT + (2 * wave + i) * 1024);
            };
            issue(0, 0);
            if (nsteps > 1) issue(1, 1);
            int slot = 0, s2 = 0;
            auto half = [&](auto H) {
                constexpr int h = decltype(H)::value;
                for (int ks = 0; ks < nk; ++ks, ++s2) {
                    if (s2 + 1 < nsteps) WF_WAIT_VM(2); else WF_WAIT_VM(0);
                    __builtin_amdgcn_s_barrier();
                    if (s2 + 2 < nsteps) issue(s2 + 2, slot == 0 ? 2 : slot - 1);
                    const unsigned ach = a_row_off + ((((unsigned)(ks * 4 + lq)) ^ (unsigned)li) << 4);       // row & 15 == li for every row tile
                    const unsigned char* Ws = Wr + slot * WF_W2_SLOT + w_ld_off;
                    bf16x8 fa[3];
#pragma unroll
                    for (int rt = 0; rt < 3; ++rt) fa[rt] = *(const bf16x8*)(lds + ach + rt * (16 * WF_NP * 2));
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const bf16x8 fb = *(const bf16x8*)(Ws + c * 1024);
#pragma unroll
                        for (int rt = 0; rt < 3; ++rt) {
                            if (WF_ABLATE & 4) asm volatile("" :: "v"(fa[rt]), "v"(fb));
                            else acc[rt][4 * h + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[rt], fb, acc[rt][4 * h + c], 0, 0, 0);
                        }
                    }
                    slot = slot == 2 ? 0 : slot + 1;
                }
            };
            half(std::integral_constant<int, 0>{});
            half(std::integral_constant<int, 1>{});
            __syncthreads();                                 // every wave is through its reads of the activations
            epilogue(l);
        }
        // ------------------------------------------------------------------ last layer: nout <= 16 f32 dot products per row
        if (tid < 4 * WF_BM) {
            const int row = tid >> 2, q = tid & 3;
            const int K = a.Klast;                            // = width of the last hidden layer (<= 512)
            const int kq = (K + 3) / 4;
            const int k0 = q * kq, k1 = (k0 + kq < K) ? k0 + kq : K;
            for (int o = 0; o < a.nout; ++o) {
                float s = 0.f;
                for (int k = k0; k < k1; ++k) {
                    const u16 hv = *(const u16*)(lds + act_off(row, k));
                    const float av = __builtin_bit_cast(float, (unsigned)hv << 16);
                    s = fmaf(av, a.Wlast[(long long)k * a.nout + o], s);       // Flux (out, in) column-major: element (o, k) at o + k nout
                }
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                if (q == 0 && row0 + row < a.S) a.out[(row0 + row) * a.out_ld + o] = wf_act_rt(a.lastact, s + a.blast[o]);
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
    switch (a.act) {
        case ISK_ACT_SIGMOID: fn = (const void*)wide_fwd_bf16_kernel<ISK_ACT_SIGMOID>; break;
        case ISK_ACT_TANH: fn = (const void*)wide_fwd_bf16_kernel<ISK_ACT_TANH>; break;
        case ISK_ACT_RELU: fn = (const void*)wide_fwd_bf16_kernel<ISK_ACT_RELU>; break;
        case ISK_ACT_SWISH: fn = (const void*)wide_fwd_bf16_kernel<ISK_ACT_SWISH>; break;
        default: fn = (const void*)wide_fwd_bf16_kernel<ISK_ACT_IDENTITY>; break;
    }
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, WF_LDS_BYTES) != hipSuccess) return ISK_EHIP;
    const long long nblk = (a.S + WF_BM - 1) / WF_BM;
    const unsigned grid = (unsigned)(nblk < ISK_NUM_CU ? nblk : ISK_NUM_CU);
    void* args[] = {(void*)&a};
    return hipLaunchKernel(fn, dim3(grid), dim3(512), args, WF_LDS_BYTES, st) == hipSuccess ? 0 : ISK_EHIP;
}
