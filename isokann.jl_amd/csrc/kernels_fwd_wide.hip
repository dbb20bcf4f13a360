// Fused bf16 forward for WIDE networks (hidden widths <= 512: C5's densenet [2000,512,512,512,1]): the Koopman forward of
// forward_mode 2 (bf16 operands, f32 accumulate -- the arithmetic BASELINE.json names for C5) as ONE kernel in which the
// activations of a 128-row block never leave the CU.  The layer-by-layer path it replaces wrote and re-read a [rows][512]
// activation buffer per layer through HBM (8.4 GB per launch at C5-shard size against 1.6 GB of bf16 features).
//
// Workgroup = 8 waves = 128 rows; wave w owns ALL 128 rows x the 64 columns [64 w, +64) of a layer's (padded) 512 outputs:
// 8 x 4 accumulator tiles of v_mfma_f32_16x16x32_bf16 = 128 registers.  The product is computed transposed -- the WEIGHT
// fragment is the MFMA's A operand, the sample fragment its B operand -- so a lane ends up with four consecutive output
// COLUMNS of one row: the next layer's contraction index, one ds_write_b64 per tile.
//   weights   never pass through LDS.  Every layer has a FRAGMENT-LINEAR bf16 image (isk_launch_wt_images, Wf): the 16 bytes
//             lane L feeds the MFMA for 16-column block nb and 32-feature step kb sit at ((kb * 32 + nb) * 64 + L) * 16, so a
//             wave's four fragments of a step are 4 contiguous KiB, read with four fully coalesced global_load_dwordx4 straight
//             into registers, FOUR steps ahead (a ring of 4 x 16 registers).  No two waves read the same bytes, nothing
//             is staged or re-read through LDS, and no barrier guards them: with the 64 x 128 wave tiles of the previous
//             kernel the W fragments were two thirds of the LDS traffic (96 KiB read + 40 KiB written per 32-feature step
//             against 1024 matrix-pipe clocks: the LDS port was as busy as the matrix pipe).
//   layer 1   the feature rows (bf16, as the featurizer writes them for this path: FeatArgs.out_bf16) in groups of 128
//             features: 128 rows x 256 bytes = 32 KiB per slot, two slots, ONE group ahead, ONE raw barrier per group.  A
//             thread requests its 4 x 16 bytes of group g + 1 with plain global_load_dwordx4 at the top of group g and
//             stores them (ds_write_b128) at its end -- registers, not LDS-DMA: see the note in front of the kernel.  A load
//             instruction covers 4 rows x 256 contiguous bytes; the 16-byte pieces of a row are XOR-swizzled with row & 15
//             through the SOURCE addresses, so the image is written linearly and the fragment reads are conflict-free.
//   layer l>1 B operand = the previous layer's activations, resident in LDS as bf16 [128][512] (16-byte chunks XORed with
//             row & 15); NO barrier inside the K-loop (waves drift apart: one wave's LDS reads run under its SIMD partner's
//             MFMAs); the outputs stay in registers until every wave has finished reading, then overwrite the activations.
//   last      nout <= 16 outputs per row: plain f32 dot products against the f32 weights, four lanes per row.
//   next block: its first feature group is requested in front of the last layer and stored behind it, into the slot that
//             lies outside the activation image; its first weight fragments are requested by the last hidden layer's final
//             steps (the weight stream is continuous) -- the block boundary exposes no round trip.
// LDS: [activations 128 KiB | feature slot 0 32 KiB]; feature slot 1 aliases the first 32 KiB of the activation image (dead
// while layer 1 runs).
#include <type_traits>

#include "isk_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define WF_BM 128
#define WF_NP 512
#define WF_ACT_BYTES (WF_BM * WF_NP * 2)                 // 131072
#define WF_A_SLOT (WF_BM * 256)                          // layer-1 feature slot: 128 rows x 128 bf16 features
#define WF_LDS_BYTES (WF_ACT_BYTES + WF_A_SLOT)          // 163840
#define WF_WSTEP (32 * 1024)                             // bytes of a layer's fragment-linear image per 32-feature step

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

// Why the feature groups are staged through registers (global_load_dwordx4 -> 16 registers -> ds_write_b128) and not moved by
// LDS-DMA, which the previous versions of this kernel used: hipcc's waitcnt insertion cannot mix the two kinds of request.
//   * every LDS access it can see behind a __builtin_amdgcn_global_load_lds waits for that DMA (it cannot know which bytes it
//     writes): the group requested a moment ago;
//   * with LDS-DMA and ordinary loads in flight together it treats vmcnt as out of order (two event types on one counter) and
//     waits for vmcnt(0) at every use of a loaded register;
//   * an LDS-DMA hidden in inline assembly is missing from its arithmetic for the weight loads: the waits come out 4 short and
//     force a feature group to have landed three steps after its request, whatever ring depth the code asks for;
//   * hiding the weight loads as well (inline assembly + hand-written waits tied to the ring registers) made the register
//     allocator spill 800 registers around the tied operands.
// With plain loads only, every wait in the K-loops is the compiler's own and exact (vmcnt(16..19) with a feature group in flight,
// vmcnt(12..15) without; lgkmcnt(3) for the fragment pairs).  The two builds run equally fast (A/B on one box).

// diagnostic builds only (timing valid, results not): -DWF_ABLATE=bits -- 1: the feature groups re-read ONE group (cache
// resident: no HBM stream), 2: every weight step re-reads the layer's first step, 4: no MFMAs (the fragment reads stay),
// 8: no feature loads at all, 16: no weight loads after a layer's first four steps, 32: no activation function (bias only)
#ifndef WF_ABLATE
#define WF_ABLATE 0
#endif

template <int ACT>
__global__ __launch_bounds__(512) void wide_fwd_bf16_kernel(WideFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // wave-uniform: addresses built from it stay in SGPRs
    const long long nblk = (a.S + WF_BM - 1) / WF_BM;
    const int ng1 = (a.Kp[0] + 127) >> 7;                                // ring groups of layer 1: 128 features (four steps) each
    const int kbytes1 = a.Kp[0] * 2;                                     // live bytes of a feature row ...
    const int xstride = a.ldx * 2;                                       // ... and its stride: rows start on 128-byte lines
    // Register budget (two waves per SIMD: 256): 128 accumulators + 64 weight-ring + 16 sample-fragment + 16 feature-staging
    // registers leave ~30 for everything else.  Addresses are therefore (uniform 64-bit base in SGPRs) + (one 32-bit lane offset) + (instruction immediate), and
    // whatever depends on the lane index is re-derived where it is used from a copy of it the compiler cannot see through
    // (WF_LANE): hoisted out of the loops, the per-instruction address registers were spilled (300 registers of scratch).
#define WF_LANE(l_) unsigned l_ = (unsigned)lane; asm volatile("" : "+v"(l_))
    // two feature slots: slot 0 (even groups) behind the activation image, slot 1 (odd groups) over its first 32 KiB
    auto slot_off = [](int s) -> unsigned { return s == 0 ? (unsigned)WF_ACT_BYTES : 0u; };

    // the weight-fragment ring: set i holds step (4 g + i) of the running layer.  Every image is padded with zero steps to a
    // multiple of FOUR steps, and the stream of (layer, step) pairs a block consumes is continuous: the group that runs a layer's
    // last four steps requests the first four of the next layer (after the last hidden layer: of layer 1, for the next block) --
    // every group is the same straight-line code, the ring never drains, and no load is conditional
    bf16x8 fw[4][4];
    const unsigned w_lane = (unsigned)lane * 16u;
    auto load_w = [&](const unsigned char* p, auto I) {            // p: uniform
        constexpr int i = decltype(I)::value;
        asm volatile("" : "+s"(p));
#pragma unroll
        for (int c = 0; c < 4; ++c) fw[i][c] = *(const __attribute__((address_space(1))) bf16x8*)(p + w_lane + c * 1024);
    };
    // feature group g of block `row0`: load q takes rows 16 wave + 4 q + lq, 16 bytes per lane -- physical piece li of a row holds
    // logical piece li ^ (row & 15) = li ^ (4 q + lq), so the image is written linearly (lane L of load q at (16 wave + 4 q) * 256
    // + 16 L) and read conflict-free.  Bytes past the (padded) row re-read its last 16 (their step multiplies zero weights); rows
    // past the end re-read the last row (never stored).
    auto load_a = [&](long long row0, int g, f32x4 (&stage)[4]) {
        if (WF_ABLATE & 8) return;
        const unsigned char* xb = (const unsigned char*)a.X + (size_t)row0 * xstride;            // uniform
        asm volatile("" : "+s"(xb));
        const int gg = (WF_ABLATE & 1) ? 0 : g;
        const long long left = a.S - 1 - row0;
        const int rmax = left < WF_BM - 1 ? (int)left : WF_BM - 1;
        WF_LANE(ln);
        const int li_ = ln & 15, lq_ = ln >> 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int r = 16 * wave + 4 * q + lq_;
            r = r < rmax ? r : rmax;
            int kb = gg * 256 + ((li_ ^ (4 * q + lq_)) << 4);
            kb = kb < kbytes1 - 16 ? kb : kbytes1 - 16;
            // (non-temporal: rows that are read once should not push the weight images out of the XCD's L2 -- A/B on one box: -1.5 %)
            stage[q] = __builtin_nontemporal_load((const __attribute__((address_space(1))) f32x4*)(xb + (unsigned)(r * xstride + kb)));
        }
    };
    auto store_a = [&](int slot, const f32x4 (&stage)[4]) {
        if (WF_ABLATE & 8) return;
        WF_LANE(ln);
        unsigned char* dst = lds + slot_off(slot) + (16 * wave) * 256 + ln * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) *(f32x4*)(dst + q * 1024) = stage[q];
    };
    // raw barrier behind this wave's own LDS stores (a __syncthreads() would also drain vmcnt: the weight ring and the staged group)
#define WF_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

    f32x4 acc[8][4];
    // (cleared in front of every K-loop, not behind the epilogues: 128 registers of zeros carried across the last layer left no
    // room for the staging registers of the next block's first feature group)
    auto zero_acc = [&]() {
#pragma unroll
        for (int rt = 0; rt < 8; ++rt)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[rt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };

    // A group of four 32-feature steps.  `abx` = LDS offset of this lane's row (li) of row tile 0 in the group's 256-byte run,
    // with the swizzle bits of step 0 merged in: piece (4 i + lq) ^ li = (4 i) ^ (lq ^ li), so step i is abx ^ 64 i; row tile rt
    // is rt * rstride further.  `wr` = the four steps that follow the ring's contents in the weight stream (this wave's 4 KiB of
    // each).  The sample fragments are read in PAIRS of row tiles, double-buffered (2 x 2 x 4 registers: the budget has no room
    // for a whole step's eight), each pair one phase ahead of the eight MFMAs that use it -- a wave's own LDS latency hides under
    // its own MFMAs, whatever its SIMD partner is doing.  The scheduling barriers pin exactly this order.
    auto group = [&](auto RS, unsigned abx, const unsigned char* wr) {
        constexpr int rstride = decltype(RS)::value;
        bf16x8 pa[2], pb[2];
        auto rd2 = [&](unsigned base, auto RT0, bf16x8 (&f)[2]) {
            constexpr int rt0 = decltype(RT0)::value;
            const unsigned b = rt0 < 4 ? base : base + 4u * rstride;      // (the immediate offset of a DS instruction is 16 bits)
            f[0] = *(const bf16x8*)(lds + b + (rt0 & 3) * rstride);
            f[1] = *(const bf16x8*)(lds + b + ((rt0 & 3) + 1) * rstride);
        };
        auto mm8 = [&](bf16x8 (&f)[2], auto RT0, auto I) {
            constexpr int rt0 = decltype(RT0)::value, i = decltype(I)::value;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (WF_ABLATE & 4) asm volatile("" :: "v"(f[t]), "v"(fw[i][c]));
                    else acc[rt0 + t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i][c], f[t], acc[rt0 + t][c], 0, 0, 0);
                }
        };
#define WF_IC(n_) std::integral_constant<int, n_>{}
#define WF_SB() __builtin_amdgcn_sched_barrier(0)
        rd2(abx, WF_IC(0), pa);
#define WF_STEP(I_)                                                                                                         \
        rd2(abx ^ (64u * I_), WF_IC(2), pb); WF_SB();                                                                       \
        mm8(pa, WF_IC(0), WF_IC(I_)); WF_SB();                                                                              \
        rd2(abx ^ (64u * I_), WF_IC(4), pa); WF_SB();                                                                       \
        mm8(pb, WF_IC(2), WF_IC(I_)); WF_SB();                                                                              \
        rd2(abx ^ (64u * I_), WF_IC(6), pb); WF_SB();                                                                       \
        mm8(pa, WF_IC(4), WF_IC(I_)); WF_SB();                                                                              \
        if (I_ < 3) rd2(abx ^ (64u * (I_ + 1)), WF_IC(0), pa);                                                              \
        WF_SB();                                                                                                            \
        mm8(pb, WF_IC(6), WF_IC(I_)); WF_SB();                                                                              \
        if (!(WF_ABLATE & 16)) load_w(wr + ((WF_ABLATE & 2) ? 0 : I_ * WF_WSTEP), WF_IC(I_));                               \
        WF_SB();
        WF_STEP(0) WF_STEP(1) WF_STEP(2) WF_STEP(3)
#undef WF_STEP
    };

    // epilogue of layer l: bias + activation, bf16, in place over the ring / the previous activations.  Element (rt, c, r) is row
    // 16 rt + li, column 64 wave + 16 c + 4 lq + r: the four r of a tile are 8 contiguous bytes of the row
    auto epilogue = [&](int l) {
        const float* bias = a.bias[l];
        const int Nl = a.N[l];
        WF_LANE(ln);
        const int li_ = ln & 15, lq_ = ln >> 4;
        float bv[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {                    // the bias loads together, from clamped addresses (no load behind a branch)
                const int col = 64 * wave + 16 * c + 4 * lq_ + r;
                bv[c][r] = bias[col < Nl ? col : Nl - 1];
                if (ACT == ISK_ACT_SIGMOID) bv[c][r] *= -1.44269504f;      // sigmoid(z + b) = 1 / (1 + 2^(z (-log2 e) + b (-log2 e))): one fma per element
            }
        const unsigned rowoff = (unsigned)(li_ * (WF_NP * 2)) + (unsigned)((lq_ & 1) << 3);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col0 = 64 * wave + 16 * c + 4 * lq_;
            const unsigned off = rowoff + ((((unsigned)(col0 >> 3)) ^ (unsigned)li_) << 4);
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // (padded columns: the accumulator is exactly 0 -- zero weight rows -- so the value is act(bias), finite, and
                    // the next layer's image has zero weights for it: no select needed)
                    if (WF_ABLATE & 32) v[r] = acc[rt][c][r] + bv[c][r];
                    else if (ACT == ISK_ACT_SIGMOID)
                        v[r] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(acc[rt][c][r], -1.44269504f, bv[c][r])));
                    else
                        v[r] = wf_act<ACT>(acc[rt][c][r] + bv[c][r]);
                }
                const bf16x4 h = __builtin_convertvector(v, bf16x4);
                *(u32x2*)(lds + off + rt * (16 * WF_NP * 2)) = __builtin_bit_cast(u32x2, h);
            }
        }
        WF_LDS_BARRIER();
    };

    const unsigned char* const wf0 = (const unsigned char*)a.Wt[0] + (size_t)wave * 4096;
    long long blk = blockIdx.x;
    if (blk < nblk) {
        f32x4 stage[4];
        load_a(blk * WF_BM, 0, stage);
        load_w(wf0, std::integral_constant<int, 0>{});
        load_w(wf0 + WF_WSTEP, std::integral_constant<int, 1>{});
        load_w(wf0 + 2 * WF_WSTEP, std::integral_constant<int, 2>{});
        load_w(wf0 + 3 * WF_WSTEP, std::integral_constant<int, 3>{});
        store_a(0, stage);
    }
    for (; blk < nblk; blk += gridDim.x) {
        const long long row0 = blk * WF_BM;
        const long long next = blk + gridDim.x;
        // ------------------------------------------------------------------ layer 1 ------------------------------------------
        {
            const unsigned char* wn = a.nhidden > 1 ? (const unsigned char*)a.Wt[1] + (size_t)wave * 4096 : wf0;
            // Barrier before group g: every wave's share of group g is in LDS (stored at the end of group g - 1; group 0: a layer
            // ago / above), and every wave is through group g - 1, whose slot this group's stores (of group g + 1) overwrite.
            // The last group is peeled: inside the loop the feature loads are unconditional, so the compiler's vmcnt arithmetic
            // for the weight ring is exact (a conditional request made it assume the smaller count -- its waits then forced the
            // feature group to have landed three steps after it was requested)
            zero_acc();
            int g = 0;
            for (; g + 1 < ng1; ++g) {
                WF_LDS_BARRIER();
                f32x4 stage[4];
                load_a(row0, g + 1, stage);
                WF_LANE(ln);
                const unsigned li_ = ln & 15, lq_ = ln >> 4;
                const unsigned abx = (slot_off(g & 1) + li_ * 256u) | ((li_ ^ lq_) << 4);
                group(std::integral_constant<int, 4096>{}, abx, wf0 + (size_t)(4 * g + 4) * WF_WSTEP);
                store_a((g + 1) & 1, stage);
            }
            {
                WF_LDS_BARRIER();
                WF_LANE(ln);
                const unsigned li_ = ln & 15, lq_ = ln >> 4;
                const unsigned abx = (slot_off(g & 1) + li_ * 256u) | ((li_ ^ lq_) << 4);
                group(std::integral_constant<int, 4096>{}, abx, wn);
            }
        }
        WF_LDS_BARRIER();                                     // every wave is through its last reads: the ring is dead
        epilogue(0);
        // ------------------------------------------------------------------ hidden layers l > 1: B operand from the activation image
        for (int l = 1; l < a.nhidden; ++l) {
            const unsigned char* wl = (const unsigned char*)a.Wt[l] + (size_t)wave * 4096;
            const unsigned char* wn = l + 1 < a.nhidden ? (const unsigned char*)a.Wt[l + 1] + (size_t)wave * 4096 : wf0;
            const int ng = (a.Kp[l] + 127) >> 7;
            zero_acc();
            for (int g = 0; g < ng; ++g) {
                WF_LANE(ln);
                const unsigned li_ = ln & 15, lq_ = ln >> 4;
                const unsigned abx = (li_ * (unsigned)(WF_NP * 2) + (unsigned)g * 256u) | ((li_ ^ lq_) << 4);
                group(std::integral_constant<int, 16 * WF_NP * 2>{}, abx, g + 1 < ng ? wl + (size_t)(4 * g + 4) * WF_WSTEP : wn);
            }
            WF_LDS_BARRIER();                                 // every wave is through its reads of the activations
            epilogue(l);
        }
        // ------------------------------------------------------------------ last layer: nout <= 16 f32 dot products per row
        // The next block's first feature group is requested in front of it and stored behind it, into the slot outside the
        // activation image: here the accumulators are dead and the staging registers cost nothing (requested earlier they were
        // spilled, i.e. waited for at once)
        f32x4 nstage[4];                                      // (its own registers: nothing is carried here from the K-loops)
        if (next < nblk) load_a(next * WF_BM, 0, nstage);
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
        if (next < nblk) store_a(0, nstage);
        WF_LDS_BARRIER();                                      // the next block's feature slot 1 overwrites the activations
    }
}

int isk_wide_fwd_ok(const WideFwdArgs& a) {
    if (a.nhidden < 1 || a.nhidden >= ISK_MAX_LAYERS || a.nout < 1 || a.nout > 16) return 0;
    for (int l = 0; l < a.nhidden; ++l)
        if (a.N[l] < 1 || a.N[l] > WF_NP || a.Kp[l] % 32 || a.Kp[l] < 32 || (l > 0 && a.Kp[l] > WF_NP)) return 0;
    return a.Klast == a.N[a.nhidden - 1] && a.Fp == a.Kp[0] && a.x_bf16 && a.ldx >= a.Kp[0] && a.ldx % 8 == 0;
}

int isk_launch_wide_fwd(const WideFwdArgs& a, hipStream_t st) {
    if (a.S <= 0) return 0;
    if (!isk_wide_fwd_ok(a)) return ISK_EUNSUPPORTED;
    const void* fn;
#define WFPICK(A_) ((const void*)wide_fwd_bf16_kernel<A_>)
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
