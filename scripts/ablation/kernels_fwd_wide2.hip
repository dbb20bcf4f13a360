// Fused bf16 forward for wide networks, TWO 64-row workgroups per CU (variant of kernels_fwd_wide.hip; round 5).
// NOT part of the library: built, parity-tested (the wide-forward tests pass on it) and measured against kernels_fwd_wide.hip in
// one call on one box -- C5 shard forward 1.127 / 1.128 ms against 1.120 / 1.123 ms: EQUAL.  Two independent workgroups per CU do not
// raise the matrix pipe's share, so the lockstep of the eight waves (barriers, epilogues reached together) is not what holds the
// 128-row kernel at 44 % of the bf16 peak.  To try it again: copy into isokann.jl_amd/csrc/, declare isk_launch_wide_fwd2 in
// isk_internal.h and call it from wide_forward (engine_layered.hip).  Two compiler findings are kept in the comments below
// (13-bit load offsets, vmcnt(0) behind every load that sits next to its use).
//
// kernels_fwd_wide.hip runs one 8-wave workgroup of 128 rows per CU: its eight waves go through a layer in lockstep (a barrier per
// 128-feature group in layer 1, one around every epilogue), so both waves of a SIMD reach their epilogues, their barriers and the
// exposed part of the feature stream's latency TOGETHER and the matrix pipe idles meanwhile (49 % busy).  Here a
// workgroup is FOUR waves and 64 rows, and two of them share a CU: one wave of each per SIMD, with nothing that couples the two --
// they drift apart, and one workgroup's epilogue / barrier / feature wait runs under the other's MFMAs.
//   wave w owns all 64 rows x the 128 columns [128 w, +128): 4 x 8 accumulator tiles (128 registers, as before), computed
//   transposed (weight fragment = A operand, sample fragment = B operand);
//   weights: the same fragment-linear images, a wave's EIGHT fragments of a 32-feature step are 8 contiguous KiB, ring of TWO steps
//   (2 x 8 x 4 = 64 registers, as before) -- both workgroups of a CU stream every image: twice the L2 -> CU weight traffic of the
//   128-row kernel (6.2 MB per 128 rows), still a fraction of what a CU can take;
//   samples: per step FOUR row-tile fragments from LDS instead of eight (half the LDS reads per MFMA);
//   LDS: [activations 64 KiB | feature slot 0 16 KiB] = 80 KiB per workgroup, two per CU; feature slot 1 aliases the first 16 KiB of
//   the activation image (dead while layer 1 runs).
// Everything else (feature groups of 128 through registers one group ahead, XOR swizzle, the continuous weight stream across layers
// and blocks, the last layer as f32 dot products) is kernels_fwd_wide.hip's.
#include <type_traits>

#include "isk_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define W2_BM 64
#define W2_NP 512
#define W2_ACT_BYTES (W2_BM * W2_NP * 2)                 // 65536
#define W2_A_SLOT (W2_BM * 256)                          // 16384
#define W2_LDS_BYTES (W2_ACT_BYTES + W2_A_SLOT)          // 81920: two workgroups per CU
#define W2_WSTEP (32 * 1024)

__device__ __forceinline__ float w2_sigmoid(float z) {
    float e = __expf(-fabsf(z));
    float r = 1.0f / (1.0f + e);
    return z >= 0.f ? r : e * r;
}
template <int ACT>
__device__ __forceinline__ float w2_act(float z) {
    if (ACT == ISK_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * z));
    if (ACT == ISK_ACT_TANH) return tanhf(z);
    if (ACT == ISK_ACT_RELU) return fmaxf(z, 0.f);
    if (ACT == ISK_ACT_SWISH) return z * w2_sigmoid(z);
    return z;
}
__device__ __forceinline__ float w2_act_rt(int act, float z) {
    switch (act) {
        case ISK_ACT_SIGMOID: return w2_sigmoid(z);
        case ISK_ACT_TANH: return tanhf(z);
        case ISK_ACT_RELU: return fmaxf(z, 0.f);
        case ISK_ACT_SWISH: return z * w2_sigmoid(z);
        default: return z;
    }
}

template <int ACT>
__global__ __launch_bounds__(256, 2) void wide_fwd2_bf16_kernel(WideFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long nblk = (a.S + W2_BM - 1) / W2_BM;
    const int ng1 = (a.Kp[0] + 127) >> 7;
    const int kbytes1 = a.Kp[0] * 2;
    const int xstride = a.ldx * 2;
#define W2_LANE(l_) unsigned l_ = (unsigned)lane; asm volatile("" : "+v"(l_))
    auto slot_off = [](int s) -> unsigned { return s == 0 ? (unsigned)W2_ACT_BYTES : 0u; };

    // weight ring: set i holds step (2 h + i) of the running layer's stream; a wave's eight fragments of a step are 8 contiguous KiB
    bf16x8 fw[2][8];
    const unsigned w_lane = (unsigned)lane * 16u;
    auto load_w = [&](const unsigned char* p, auto I) {
        constexpr int i = decltype(I)::value;
        // two uniform bases 4 KiB apart: the immediate offset of a global load is 13 bits signed, and a fragment at +4096 .. +7168
        // from ONE base made the compiler build 64-bit addresses in registers -- the pair that then spilled a staging register
        // INSIDE the layer-1 loop (its reload waits for vmcnt(0): the weight ring drained once per feature group)
        const unsigned char* p2 = p + 4096;
        asm volatile("" : "+s"(p), "+s"(p2));
#pragma unroll
        for (int c = 0; c < 4; ++c) fw[i][c] = *(const __attribute__((address_space(1))) bf16x8*)(p + w_lane + c * 1024);
#pragma unroll
        for (int c = 0; c < 4; ++c) fw[i][4 + c] = *(const __attribute__((address_space(1))) bf16x8*)(p2 + w_lane + c * 1024);
    };
    // HALF h of feature group g (loads q = 2 h, 2 h + 1): the group is staged in two halves of 8 registers, each requested two
    // steps before it is stored (16 staging registers beside 128 accumulators + 64 ring + 16 fragment registers left no room for
    // the loop's addresses: one of them was spilled and reloaded INSIDE the layer-1 loop, a vmcnt(0) drain per group)
    auto load_a = [&](long long row0, int g, int h, f32x4 (&stage)[2]) {
        const unsigned char* xb = (const unsigned char*)a.X + (size_t)row0 * xstride;
        asm volatile("" : "+s"(xb));
        const long long left = a.S - 1 - row0;
        const int rmax = left < W2_BM - 1 ? (int)left : W2_BM - 1;
        W2_LANE(ln);
        const int li_ = ln & 15, lq_ = ln >> 4;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int q = 2 * h + qq;
            int r = 16 * wave + 4 * q + lq_;
            r = r < rmax ? r : rmax;
            int kb = g * 256 + ((li_ ^ (4 * q + lq_)) << 4);
            kb = kb < kbytes1 - 16 ? kb : kbytes1 - 16;
            stage[qq] = __builtin_nontemporal_load((const __attribute__((address_space(1))) f32x4*)(xb + (unsigned)(r * xstride + kb)));
        }
    };
    auto store_a = [&](int slot, int h, const f32x4 (&stage)[2]) {
        W2_LANE(ln);
        unsigned char* dst = lds + slot_off(slot) + (16 * wave) * 256 + ln * 16 + h * 2048;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) *(f32x4*)(dst + qq * 1024) = stage[qq];
    };
#define W2_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

    f32x4 acc[4][8];
    auto zero_acc = [&]() {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[rt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };

    // A group of four 32-feature steps.  abx: LDS offset of this lane's row (li) of row tile 0 in the group's 256-byte run with the
    // swizzle bits of step 0 merged in (step i is abx ^ 64 i; row tile rt is rt * rstride further).  wcur: this wave's fragments of
    // the group's first step in the weight stream, wnext: those of the NEXT group's first step (same layer, next layer, or layer 1 of
    // the next block): the ring holds steps 0, 1 at entry; step I is followed by the load of step I + 2.
    auto group = [&](auto RS, unsigned abx, const unsigned char* wcur, const unsigned char* wnext, auto&& mid) {
        constexpr int rstride = decltype(RS)::value;
        bf16x8 pa[2], pb[2];
        auto rd2 = [&](unsigned base, auto RT0, bf16x8 (&f)[2]) {
            constexpr int rt0 = decltype(RT0)::value;
            f[0] = *(const bf16x8*)(lds + base + rt0 * rstride);
            f[1] = *(const bf16x8*)(lds + base + (rt0 + 1) * rstride);
        };
        auto mm16 = [&](bf16x8 (&f)[2], auto RT0, auto I) {
            constexpr int rt0 = decltype(RT0)::value, i = decltype(I)::value;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[rt0 + t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i][c], f[t], acc[rt0 + t][c], 0, 0, 0);
        };
#define W2_IC(n_) std::integral_constant<int, n_>{}
#define W2_SB() __builtin_amdgcn_sched_barrier(0)
        rd2(abx, W2_IC(0), pa);
#define W2_STEP(I_, WP_)                                                                                                    \
        rd2(abx ^ (64u * I_), W2_IC(2), pb); W2_SB();                                                                       \
        mm16(pa, W2_IC(0), W2_IC(I_ & 1)); W2_SB();                                                                         \
        if (I_ < 3) rd2(abx ^ (64u * (I_ + 1)), W2_IC(0), pa);                                                              \
        W2_SB();                                                                                                            \
        mm16(pb, W2_IC(2), W2_IC(I_ & 1)); W2_SB();                                                                         \
        load_w(WP_, W2_IC(I_ & 1));                                                                                         \
        W2_SB();
        W2_STEP(0, wcur + 2 * W2_WSTEP) W2_STEP(1, wcur + 3 * W2_WSTEP)
        mid();
        W2_SB();
        W2_STEP(2, wnext) W2_STEP(3, wnext + W2_WSTEP)
#undef W2_STEP
    };

    // epilogue of layer l: bias + activation, bf16, in place.  Element (rt, c, r): row 16 rt + li, column 128 wave + 16 c + 4 lq + r
    auto epilogue = [&](int l) {
        const float* bias = a.bias[l];
        const int Nl = a.N[l];
        W2_LANE(ln);
        const int li_ = ln & 15, lq_ = ln >> 4;
        // every bias value requested up front, from clamped addresses (requested tile by tile each load was waited for with
        // vmcnt(0): 32 dependent round trips per epilogue); the weight ring is dead here, its registers hold them
        float bvv[8][4];
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = 128 * wave + 16 * c + 4 * lq_ + r;
                bvv[c][r] = bias[col < Nl ? col : Nl - 1];
            }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned rowoff = (unsigned)(li_ * (W2_NP * 2)) + (unsigned)((lq_ & 1) << 3);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int col0 = 128 * wave + 16 * c + 4 * lq_;
            float bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = ACT == ISK_ACT_SIGMOID ? bvv[c][r] * -1.44269504f : bvv[c][r];
            const unsigned off = rowoff + ((((unsigned)(col0 >> 3)) ^ (unsigned)li_) << 4);
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (ACT == ISK_ACT_SIGMOID)
                        v[r] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(acc[rt][c][r], -1.44269504f, bv[r])));
                    else
                        v[r] = w2_act<ACT>(acc[rt][c][r] + bv[r]);
                }
                const bf16x4 h = __builtin_convertvector(v, bf16x4);
                *(u32x2*)(lds + off + rt * (16 * W2_NP * 2)) = __builtin_bit_cast(u32x2, h);
            }
        }
        W2_LDS_BARRIER();
    };

    const unsigned char* const wf0 = (const unsigned char*)a.Wt[0] + (size_t)wave * 8192;
    long long blk = blockIdx.x;
    if (blk < nblk) {
        f32x4 s0[2], s1[2];
        load_a(blk * W2_BM, 0, 0, s0);
        load_a(blk * W2_BM, 0, 1, s1);
        load_w(wf0, std::integral_constant<int, 0>{});
        load_w(wf0 + W2_WSTEP, std::integral_constant<int, 1>{});
        store_a(0, 0, s0);
        store_a(0, 1, s1);
    }
    for (; blk < nblk; blk += gridDim.x) {
        const long long row0 = blk * W2_BM;
        const long long next = blk + gridDim.x;
        // ------------------------------------------------------------------ layer 1
        {
            const unsigned char* wn = a.nhidden > 1 ? (const unsigned char*)a.Wt[1] + (size_t)wave * 8192 : wf0;
            zero_acc();
            int g = 0;
            for (; g + 1 < ng1; ++g) {
                W2_LDS_BARRIER();
                f32x4 stage[2];
                load_a(row0, g + 1, 0, stage);
                W2_LANE(ln);
                const unsigned li_ = ln & 15, lq_ = ln >> 4;
                const unsigned abx = (slot_off(g & 1) + li_ * 256u) | ((li_ ^ lq_) << 4);
                const unsigned char* wc = wf0 + (size_t)(4 * g) * W2_WSTEP;
                group(std::integral_constant<int, 4096>{}, abx, wc, wc + 4 * W2_WSTEP, [&]() {
                    store_a((g + 1) & 1, 0, stage);
                    load_a(row0, g + 1, 1, stage);
                });
                store_a((g + 1) & 1, 1, stage);
            }
            {
                W2_LDS_BARRIER();
                W2_LANE(ln);
                const unsigned li_ = ln & 15, lq_ = ln >> 4;
                const unsigned abx = (slot_off(g & 1) + li_ * 256u) | ((li_ ^ lq_) << 4);
                group(std::integral_constant<int, 4096>{}, abx, wf0 + (size_t)(4 * g) * W2_WSTEP, wn, []() {});
            }
        }
        W2_LDS_BARRIER();
        epilogue(0);
        // ------------------------------------------------------------------ hidden layers l > 1
        for (int l = 1; l < a.nhidden; ++l) {
            const unsigned char* wl = (const unsigned char*)a.Wt[l] + (size_t)wave * 8192;
            const unsigned char* wn = l + 1 < a.nhidden ? (const unsigned char*)a.Wt[l + 1] + (size_t)wave * 8192 : wf0;
            const int ng = (a.Kp[l] + 127) >> 7;
            zero_acc();
            for (int g = 0; g < ng; ++g) {
                W2_LANE(ln);
                const unsigned li_ = ln & 15, lq_ = ln >> 4;
                const unsigned abx = (li_ * (unsigned)(W2_NP * 2) + (unsigned)g * 256u) | ((li_ ^ lq_) << 4);
                const unsigned char* wc = wl + (size_t)(4 * g) * W2_WSTEP;
                group(std::integral_constant<int, 16 * W2_NP * 2>{}, abx, wc, g + 1 < ng ? wc + 4 * W2_WSTEP : wn, []() {});
            }
            W2_LDS_BARRIER();
            epilogue(l);
        }
        // ------------------------------------------------------------------ last layer: nout <= 16 f32 dot products per row
        f32x4 ns0[2], ns1[2];
        if (next < nblk) { load_a(next * W2_BM, 0, 0, ns0); load_a(next * W2_BM, 0, 1, ns1); }
        {
            const int row = tid >> 2, q = tid & 3;
            const int K = a.Klast;
            const int nch = (K + 7) >> 3;
            const unsigned char* arow = lds + row * (W2_NP * 2);
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
                if (q == 0 && row0 + row < a.S) a.out[(row0 + row) * a.out_ld] = w2_act_rt(a.lastact, s + a.blast[0]);
            } else {
                for (int o = 0; o < a.nout; ++o) {
                    float s = 0.f;
#pragma unroll 1
                    for (int c = q; c < nch; c += 4) {
                        const bf16x8 av = *(const bf16x8*)(arow + ((c ^ (row & 15)) << 4));
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int k = 8 * c + j;
                            if (k < K) s = fmaf((float)av[j], a.Wlast[(long long)k * a.nout + o], s);
                        }
                    }
                    s += __shfl_xor(s, 1, 64);
                    s += __shfl_xor(s, 2, 64);
                    if (q == 0 && row0 + row < a.S) a.out[(row0 + row) * a.out_ld + o] = w2_act_rt(a.lastact, s + a.blast[o]);
                }
            }
        }
        if (next < nblk) { store_a(0, 0, ns0); store_a(0, 1, ns1); }
        W2_LDS_BARRIER();
    }
}

int isk_launch_wide_fwd2(const WideFwdArgs& a, hipStream_t st) {
    if (a.S <= 0) return 0;
    if (!isk_wide_fwd_ok(a)) return ISK_EUNSUPPORTED;
    const void* fn;
#define W2PICK(A_) ((const void*)wide_fwd2_bf16_kernel<A_>)
    switch (a.act) {
        case ISK_ACT_SIGMOID: fn = W2PICK(ISK_ACT_SIGMOID); break;
        case ISK_ACT_TANH: fn = W2PICK(ISK_ACT_TANH); break;
        case ISK_ACT_RELU: fn = W2PICK(ISK_ACT_RELU); break;
        case ISK_ACT_SWISH: fn = W2PICK(ISK_ACT_SWISH); break;
        default: fn = W2PICK(ISK_ACT_IDENTITY); break;
    }
#undef W2PICK
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS_BYTES) != hipSuccess) return ISK_EHIP;
    const long long nblk = (a.S + W2_BM - 1) / W2_BM;
    const unsigned grid = (unsigned)(nblk < 2 * ISK_NUM_CU ? nblk : 2 * ISK_NUM_CU);
    void* args[] = {(void*)&a};
    return hipLaunchKernel(fn, dim3(grid), dim3(256), args, W2_LDS_BYTES, st) == hipSuccess ? 0 : ISK_EHIP;
}
