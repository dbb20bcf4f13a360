// Device helpers shared by the forward kernels (f32 MFMA and split-bf16 variants).
#pragma once
#include "isk_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WAVE 64

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// the same sum over the 64 lanes without LDS traffic: DPP row shifts (inclusive scan inside each row of 16), then
// row_bcast:15 / row_bcast:31 carry the row totals up; lane 63 holds the total, read back as a wave-uniform scalar.
// 8 VALU instructions instead of 6 ds_bpermute round trips (used where a wave runs alone on its SIMD).
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define ISK_DPP_ADD(ctrl, rmask)                                                                                   \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
    ISK_DPP_ADD(0x111, 0xf);      // row_shr:1
    ISK_DPP_ADD(0x112, 0xf);      // row_shr:2
    ISK_DPP_ADD(0x114, 0xf);      // row_shr:4
    ISK_DPP_ADD(0x118, 0xf);      // row_shr:8
    ISK_DPP_ADD(0x142, 0xa);      // row_bcast:15 into rows 1 and 3
    ISK_DPP_ADD(0x143, 0xc);      // row_bcast:31 into rows 2 and 3
#undef ISK_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// power-of-two scale that brings a row (or tensor) maximum m into [2^14, 2^15): the largest f16 exponent range that cannot
// overflow.  sc = 2^(141 - ex), inv = 2^(ex - 141) with ex the biased exponent of m; zero / tiny / non-finite maxima keep 1.
__host__ __device__ __forceinline__ void isk_f16_scale(float m, float& sc, float& inv) {
    unsigned ub;
    __builtin_memcpy(&ub, &m, 4);
    const int ex = (int)((ub >> 23) & 0xffu);
    sc = 1.f; inv = 1.f;
    if (ex >= 15 && ex != 255) {
        const unsigned sb = (unsigned)(268 - ex) << 23, ib = (unsigned)(ex - 14) << 23;
        __builtin_memcpy(&sc, &sb, 4);
        __builtin_memcpy(&inv, &ib, 4);
    }
}

__device__ __forceinline__ float act_apply(int act, float z) {
    switch (act) {
        case ISK_ACT_SIGMOID: {
            // NNlib sigmoid_fast semantics: overflow-free 1/(1+exp(-z))
            float e = __expf(-fabsf(z));
            float r = 1.0f / (1.0f + e);
            return z >= 0.f ? r : e * r;
        }
        case ISK_ACT_TANH: return tanhf(z);
        case ISK_ACT_RELU: return fmaxf(z, 0.f);
        case ISK_ACT_SWISH: {
            float e = __expf(-fabsf(z));
            float r = 1.0f / (1.0f + e);
            return z * (z >= 0.f ? r : e * r);
        }
        default: return z;
    }
}


// epilogue + tail layers for one 16-row unit whose layer-1 pre-activations sit in acc[NT]
// DENSE (compile time): the dense-block mode of the layered path; kept out of the fused instantiations (and the tail
// code out of the dense ones) because the split-bf16 kernel sits at the 256-register limit and every extra path spills
template <int NT, bool DENSE = false>
__device__ __forceinline__ void unit_tail(const MlpFwdArgs& a, f32x4 (&acc)[NT], long long row_base, float* scrA,
                                          float* scrB, const float* tailw, int lane, int col_shift = 0) {
    const int li = lane & 15, lq = lane >> 4;
    int stride = a.stride_a;
    if (DENSE) {
        // one layer of the layered path: straight from the accumulators (row = 4 lq + r, column = 16 t + li) to global
        // memory, 64 contiguous bytes per row and instruction -- no LDS transpose.  Every padded column of this block
        // gets act(z) (finite, so the next layer's zero weight rows annihilate it).
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = t * 16 + li;
            const float bcol = tailw[a.bias_lds0 + col];
            const bool colok = a.out_col0 + col_shift + col < a.out_ld;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long grow = row_base + lq * 4 + r;
                if (colok && grow < a.S) a.out[grow * a.out_ld + a.out_col0 + col_shift + col] = act_apply(a.act1, acc[t][r] + bcol);
            }
        }
        return;
    }
    const bool fast_sig = (a.act1 == ISK_ACT_SIGMOID) && a.ntail > 0;    // wave-uniform
    // accumulator layout (row = 4*lq + r, col = 16*t + li) -> LDS rows; bias (and, on the common
    // sigmoid path, the activation) applied on the way, in registers
    if (fast_sig) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = t * 16 + li;
            // 1 / (1 + 2^(-z log2 e)): one fma, v_exp_f32, one add, v_rcp_f32 -- the form needs no overflow guard in f32
            // (2^(+big) = inf -> 1/inf = 0, 2^(-big) = 0 -> 1) and is 40 instead of 60 issue cycles per element of the
            // 72 a lane holds per pass (the two transcendentals are quarter rate)
            const float bneg = tailw[a.bias_lds0 + col] * -1.4426950408889634f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(acc[t][r], -1.4426950408889634f, bneg));
                scrA[(lq * 4 + r) * stride + col] = __builtin_amdgcn_rcpf(1.0f + e);
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = t * 16 + li;
            const float bcol = tailw[a.bias_lds0 + col];
#pragma unroll
            for (int r = 0; r < 4; ++r) scrA[(lq * 4 + r) * stride + col] = acc[t][r] + bcol;
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int H1p = NT * 16;
    if (a.ntail == 0) {
        for (int row = lq; row < 16; row += 4) {
            const long long grow = row_base + row;
            if (li < a.nout && grow < a.S) a.out[grow * a.nout + li] = act_apply(a.act1, scrA[row * stride + li]);
        }
        // nout > 16 with a single layer
        for (int col = 16 + li; col < a.nout; col += 16)
            for (int row = lq; row < 16; row += 4) {
                const long long grow = row_base + row;
                if (grow < a.S) a.out[grow * a.nout + col] = act_apply(a.act1, scrA[row * stride + col]);
            }
        __builtin_amdgcn_wave_barrier();
        return;
    }
    if (!fast_sig) {
        for (int row = 0; row < 16; ++row)
            for (int col = lane; col < H1p; col += WAVE) scrA[row * stride + col] = act_apply(a.act1, scrA[row * stride + col]);
        __builtin_amdgcn_wave_barrier();
    }
    float* in = scrA;
    float* ob = scrB;
    int ostride = a.stride_b;
    int boff = a.bias_lds0 + NT * 16;
    for (int l = 0; l < a.ntail; ++l) {
        const TailLayer L = a.tail[l];
        const bool last = (l == a.ntail - 1);
        const f32x4* Wp = (const f32x4*)(tailw + L.lds_off) + lane;     // staged once per workgroup
        for (int t2 = 0; t2 < L.ntiles; ++t2) {
            // two independent accumulation chains (even / odd k-chunks) hide the MFMA and LDS latency
            f32x4 z = {0.f, 0.f, 0.f, 0.f}, zb = {0.f, 0.f, 0.f, 0.f};
            int c = 0;
            for (; c + 1 < L.kchunks; c += 2) {
                const f32x4 xa = *(const f32x4*)(in + li * stride + c * 16 + lq * 4);
                const f32x4 wf = Wp[(long long)(c * L.ntiles + t2) * 64];
                const f32x4 xb = *(const f32x4*)(in + li * stride + (c + 1) * 16 + lq * 4);
                const f32x4 wg = Wp[(long long)((c + 1) * L.ntiles + t2) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    z = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[j], wf[j], z, 0, 0, 0);
                    zb = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[j], wg[j], zb, 0, 0, 0);
                }
            }
            if (c < L.kchunks) {
                const f32x4 xa = *(const f32x4*)(in + li * stride + c * 16 + lq * 4);
                const f32x4 wf = Wp[(long long)(c * L.ntiles + t2) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j) z = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[j], wf[j], z, 0, 0, 0);
            }
            z += zb;
            const int col = t2 * 16 + li;
            const float bcol = tailw[boff + col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = act_apply(L.act, z[r] + bcol);
                if (last) {
                    const long long grow = row_base + lq * 4 + r;
                    if (col < a.nout && grow < a.S) a.out[grow * a.nout + col] = v;
                } else {
                    ob[(lq * 4 + r) * ostride + col] = v;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        float* tmp = in; in = ob; ob = tmp;
        const int ts = stride; stride = ostride; ostride = ts;
        boff += L.ntiles * 16;
    }
}


// stage the (small) fragment-linear tail weights in LDS once per workgroup
__device__ __forceinline__ void stage_tail_weights(const MlpFwdArgs& a, float* smem, int bias_shift = 0) {
    for (int l = 0; l < a.ntail; ++l) {
        const int n4 = a.tail[l].kchunks * a.tail[l].ntiles * 64;
        const f32x4* src = (const f32x4*)a.tail[l].Wt;
        f32x4* dst = (f32x4*)(smem + a.tail[l].lds_off);
        for (int i = threadIdx.x; i < n4; i += blockDim.x) dst[i] = src[i];
    }
    // bias block: [bias1][tail biases...]
    float* bl = smem + a.bias_lds0;
    for (int i = threadIdx.x; i < a.nt1 * 16; i += blockDim.x) bl[i] = a.bias1[bias_shift + i];
    int off = a.nt1 * 16;
    for (int l = 0; l < a.ntail; ++l) {
        const int n = a.tail[l].ntiles * 16;
        for (int i = threadIdx.x; i < n; i += blockDim.x) bl[off + i] = a.tail[l].bias[i];
        off += n;
    }
}
