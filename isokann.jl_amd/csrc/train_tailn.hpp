// train_tail for NARROW tails (shared device code: train_tailn_kernel and the persistent epoch kernel)
// train_tail for NARROW tails (gfx950): the default pairnet tails -- C3 [.,142,12,3], C2 [.,38,6,1], C1 [.,2,1,1] -- have a
// first hidden width of at most 160 and every later layer at most 16 wide.  The generic train_tail_kernel (kernels_train.hip)
// stages activations, pre-activations and weights in LDS and walks nine workgroup barriers per 16 samples with MFMA operands
// fetched element by element from LDS: 11 us in-kernel for about 50 MFMAs of arithmetic.  Here the whole tail of a 16-sample
// unit lives in REGISTERS, in one fixed lane mapping, and the f32 MFMA is used so that every product comes out in the layout
// the next one consumes:
//
//   lane = (c, q), c = lane % 16, q = lane / 16.   "row-lane" image RL of a 16 x 16 tile T: lane (c, q) holds T[row c][4q + r],
//   r = 0..3 (an f32x4).  "col-lane" image CL: lane (c, q) holds T[row 4q + r][col c].
//
//   v_mfma_f32_16x16x4_f32:  D[i][j] += sum_k A[i][k] B[k][j];  A: lane (c, q) = A[c][q];  B: lane (c, q) = B[q][c];
//   accumulator: lane (c, q) = D[4q + r][c]  -- i.e. D in CL, or D' in RL.
//
//   forward   Z_l' = W_l A_{l-1}':  A operand = W_l[n = c][k], B operand = RL of A_{l-1}  ->  accumulator = RL of Z_l
//             (the contraction index of step r is k = 16 t + 4 q + r: any bijection serves as long as both operands use it)
//   delta     delta_{l-1}' = W_l' delta_l':  A = W_l[n = 4q + r][k = 16 t + c], B = RL of delta_l  ->  RL of delta_{l-1}
//   dW_l      = delta_l' A_{l-1} (contraction over the 16 samples): A = CL of delta_l, B = CL of A_{l-1}
//             ->  accumulator lane (c, q) = dW_l[n = 4q + r][k = 16 t + c]
//
// so the only re-layouts are RL -> CL transposes of 16 x 16 tiles for the dW products (through a wave-private LDS tile, no
// workgroup barrier).  A unit is a workgroup of FOUR waves that split the layer-1 tiles (t = wave, wave + 4, ...): partial
// Z_2 tiles are summed through LDS behind the ONE barrier of the forward pass, everything narrower than 16 is computed
// redundantly by the four waves (it is a handful of MFMAs), and the dW_2 / delta_1 tiles are again split by wave.  A second
// barrier orders the delta_1 image (written to LDS in isk_delta1_pos order) before its coalesced store.
// Interface, partial-vector layout and results are those of train_tail_kernel (src/iso.jl:183-191 of the reference).
#pragma once
#include "isk_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TN_WAVES 4
#define TN_TPW 3                    /* layer-1 tiles per wave: nt1 <= 12 */
#define TN_TS 20                    /* row stride (floats) of a transpose tile: 16-byte aligned rows, spread over the banks */

__device__ __forceinline__ float tn_sigmoid(float z) {
    const float e = __expf(-fabsf(z));
    const float r = 1.0f / (1.0f + e);                             // (IEEE division: the training path is held to the oracle at fp32 rounding)
    return z >= 0.f ? r : e * r;
}
// four values per call, called UNCONDITIONALLY (masks are applied to the results): a call under a lane mask costs a divergent
// branch of its own, and the first version of this kernel had 52 of them
__device__ __attribute__((noinline)) f32x4 tn_act4(int act, f32x4 z) {
    f32x4 a;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        switch (act) {
            case ISK_ACT_SIGMOID: a[r] = tn_sigmoid(z[r]); break;
            case ISK_ACT_TANH: a[r] = tanhf(z[r]); break;
            case ISK_ACT_RELU: a[r] = fmaxf(z[r], 0.f); break;
            case ISK_ACT_SWISH: a[r] = z[r] * tn_sigmoid(z[r]); break;
            default: a[r] = z[r]; break;
        }
    }
    return a;
}
__device__ __attribute__((noinline)) f32x4 tn_dact4(int act, f32x4 z, f32x4 a) {
    f32x4 d;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        switch (act) {
            case ISK_ACT_SIGMOID: d[r] = a[r] * (1.f - a[r]); break;
            case ISK_ACT_TANH: d[r] = 1.f - a[r] * a[r]; break;
            case ISK_ACT_RELU: d[r] = z[r] > 0.f ? 1.f : 0.f; break;
            case ISK_ACT_SWISH: { const float sg = tn_sigmoid(z[r]); d[r] = sg + a[r] * (1.f - sg); } break;
            default: d[r] = 1.f; break;
        }
    }
    return d;
}
// SIG: sigmoid hidden layers and an identity last layer (pairnet's defaults, src/models.jl:65) inlined, no calls
template <bool SIG>
__device__ __forceinline__ f32x4 tn_hidden_act(int act, f32x4 z) {
    if (SIG) { f32x4 a; for (int r = 0; r < 4; ++r) a[r] = tn_sigmoid(z[r]); return a; }
    return tn_act4(act, z);
}
template <bool SIG>
__device__ __forceinline__ f32x4 tn_hidden_dact(int act, f32x4 z, f32x4 a) {
    if (SIG) { f32x4 d; for (int r = 0; r < 4; ++r) d[r] = a[r] * (1.f - a[r]); return d; }
    return tn_dact4(act, z, a);
}
__device__ __forceinline__ f32x4 tn_mask(f32x4 v, const bool (&ok)[4]) {
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = ok[r] ? v[r] : 0.f;
    return o;
}

// RL -> CL of one 16 x 16 tile through a wave-private LDS tile.  The LDS pipe serves a wave's accesses in order; the fences
// keep the compiler from moving the reads (other lanes' data) above the writes.
__device__ __forceinline__ f32x4 tn_transpose(float* tile, f32x4 v, int c, int q) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // earlier reads of this tile are done
    *(f32x4*)(tile + c * TN_TS + 4 * q) = v;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = tile[(4 * q + r) * TN_TS + c];
    return o;
}

// buffer accesses: one instruction each with a 32-bit per-lane byte offset, and a load past the end returns 0 -- which is how
// every padding element (output n >= N, input k >= K) is masked: its offset is -1
typedef unsigned tn_u32x4 __attribute__((ext_vector_type(4)));


// where a unit's inputs come from and its results go.  EP = inside the persistent epoch kernel (kernels_train_epoch.hip): z1
// arrives as an LDS image (the K-slice partials were summed by the whole workgroup), the small parameters are read and every
// result is stored WRITE-THROUGH (sc1: they are handed to other workgroups inside the launch), delta_1 keeps its natural column
// order.  Otherwise (train_tailn_kernel): K-slice partials from train_fwd1n, plain accesses, delta_1 in isk_delta1_pos order.
struct TailNIO {
    const float* theta;
    const float* target;            // [N][nout] (or the cotangent rows, cot_mode)
    const float* lossw;
    int cot_mode;
    const float* a1; long long zstride; int KS;      // !EP: partials [KS][rows][H1p]
    const float* z1img;             // EP: LDS image [16][H1p] of sum_ks partials (bias not yet added)
    const float* wlds;              // EP: LDS copy of theta[off_b0 .. nparams): the layer-1 bias and every tail parameter
    float* delta1;                  // [rows][H1p]
    float* part_out;                // this unit's partial vector [PS]
    float inv_div;
};

template <int AUX>
__device__ __forceinline__ float tn_ldw(__amdgpu_buffer_rsrc_t r, bool ok, int idx) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, ok ? idx * 4 : -1, 0, AUX));
}
template <int AUX>
__device__ __forceinline__ void tn_stw(__amdgpu_buffer_rsrc_t r, int idx, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, idx * 4, 0, AUX);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tn_rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}

// One 16-sample unit on NW waves (all of them call; unit / B / prow: this lane's sample row c = lane % 16).
// s_zred [NW][256], s_tile [NW][16 * TN_TS], s_d1 [16 * H1p] (!EP only).  Returns the unit's loss (every lane).
// ZSC1: the K-slice partials were stored by other workgroups of THIS launch (kernels_train_fwdtail.hip): sc1 loads
template <int L, bool SIG, int NW, bool EP, bool ZSC1 = false>
__device__ __forceinline__ float tailn_unit(const TailNDesc& d, const TailNIO& io, int unit, int B, int prow, float* s_zred, float* s_tile_all,
                                            float* s_d1, int wv, int ln) {
    constexpr int TPW = (12 + NW - 1) / NW < 2 ? 2 : (12 + NW - 1) / NW;      // layer-1 tiles per wave (nt <= 10)
    constexpr int AUX = EP ? 16 : 0;                                          // sc1: bypass this CU's L1 / write through
    const int c = ln & 15, q = ln >> 4;
    const int H1 = d.H1, H1p = d.H1p, nt = d.nt, nout = d.nout;
    float* const tile = s_tile_all + wv * 16 * TN_TS;
    const int row_g = unit * 16 + c;
    const bool rvalid = row_g < B;
    const TailNLay y0 = d.lay[0];
    // ---- every load of the unit is requested here
    // a parameter block [off, off + n) of theta; element idx when `ok`, else (and past the block's end) 0.  EP: from the LDS copy
    auto ldw = [&](long long off, int n, bool ok, int idx) -> float {
        if (EP) return (ok && (unsigned)idx < (unsigned)n) ? io.wlds[(int)(off - d.off_b0) + idx] : 0.f;
        return tn_ldw<0>(tn_rsrc(io.theta + off, n * 4), ok, idx);
    };
    f32x4 zsum[TPW], b1v[TPW], wf0[TPW], wb0[TPW];
    if (!EP) {
        const __amdgpu_buffer_rsrc_t rz = tn_rsrc(io.a1, 0x7fffffff);
        const int zs4 = (int)(io.zstride * 4);
        f32x4 zp[TPW][ISK_FWD1_MAXKS];
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
            const int t = wv + NW * ti, tt = t < nt ? t : nt - 1;     // absent tiles re-read the last one (never used)
            const int zoff = (row_g * H1p + 16 * tt + 4 * q) * 4;
#pragma unroll
            for (int ks = 0; ks < ISK_FWD1_MAXKS; ++ks)
                zp[ti][ks] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, zoff, (ks < io.KS ? ks : 0) * zs4, ZSC1 ? 16 : 0));
        }
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
            // (the K-slice partials are summed in double: one float rounding for the whole 1728-term product)
            double z0 = zp[ti][0][0], z1d = zp[ti][0][1], z2 = zp[ti][0][2], z3 = zp[ti][0][3];
#pragma unroll
            for (int ks = 1; ks < ISK_FWD1_MAXKS; ++ks)
                if (ks < io.KS) { z0 += (double)zp[ti][ks][0]; z1d += (double)zp[ti][ks][1]; z2 += (double)zp[ti][ks][2]; z3 += (double)zp[ti][ks][3]; }
            zsum[ti] = (f32x4){(float)z0, (float)z1d, (float)z2, (float)z3};
        }
    } else {
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
            const int t = wv + NW * ti, tt = t < nt ? t : nt - 1;
            zsum[ti] = *(const f32x4*)(io.z1img + c * H1p + 16 * tt + 4 * q);
        }
    }
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int t = wv + NW * ti, tt = t < nt ? t : nt - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * tt + 4 * q + r;                     // this lane's layer-1 output column
            b1v[ti][r] = ldw(d.off_b0, H1, true, k);               // (k >= H1 is past the end: 0)
            wf0[ti][r] = ldw(y0.gW, y0.K * y0.N, c < y0.N, c + k * y0.N);                                  // W_2[n = c][k]   (k >= H1: past the end)
            wb0[ti][r] = ldw(y0.gW, y0.K * y0.N, 4 * q + r < y0.N, 4 * q + r + (16 * tt + c) * y0.N);      // W_2[n = 4q + r][k = 16 t + c]
        }
    }
    f32x4 wfl[TN_MAXL], wbl[TN_MAXL], bl[TN_MAXL];                 // deeper layers: one tile each; bl also for layer 0
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const TailNLay y = d.lay[l];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = 4 * q + r;
            bl[l][r] = ldw(y.gb, y.N, true, n);
            if (l > 0) {
                wfl[l][r] = ldw(y.gW, y.K * y.N, c < y.N && n < y.K, c + n * y.N);         // W[n = c][k = 4q + r]
                wbl[l][r] = ldw(y.gW, y.K * y.N, n < y.N && c < y.K, n + c * y.N);         // W[n = 4q + r][k = c]
            }
        }
    }
    f32x4 tgt, lw;
    {
        const __amdgpu_buffer_rsrc_t rt = tn_rsrc(io.target, 0x7fffffff);
        const __amdgpu_buffer_rsrc_t rl = tn_rsrc(io.lossw, io.cot_mode ? 0 : nout * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 4 * q + r;
            tgt[r] = tn_ldw<0>(rt, o < nout, prow * nout + o);
            lw[r] = tn_ldw<0>(rl, true, o);
        }
    }
    bool okn[TN_MAXL][4];                                           // output 4q + r exists in tail layer l
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int r = 0; r < 4; ++r) okn[l][r] = 4 * q + r < d.lay[l].N;

    // ---- layer 1: z1 = b1 + sum of the K-slice partials, a1 = act(z1); RL images of this wave's tiles.
    // no masks: a padding column (>= H1) has z = 0 exactly (its layer-1 weights are 0, its bias reads past the end) and every
    // weight that multiplies its activation -- forward, delta and dW alike -- is a past-the-end load, i.e. 0
    f32x4 z1[TPW], a1[TPW], a1c[TPW];
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        z1[ti] = zsum[ti] + b1v[ti];
        a1[ti] = tn_hidden_act<SIG>(d.act0, z1[ti]);
    }
    // ---- first tail layer: partial Z' over this wave's tiles, summed across the waves (fixed order: every wave gets the same bits)
    {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
            if (wv + NW * ti < nt) {                                // wave-uniform
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf0[ti][r], a1[ti][r], acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf0[ti][r], a1[ti][r], acc0, 0, 0, 0);
                }
            }
        }
        *(f32x4*)(s_zred + wv * 256 + ln * 4) = acc0 + acc1;
    }
    // the CL images of a1 (for dW of the first tail layer) are taken while the other waves arrive at the barrier
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) a1c[ti] = tn_transpose(tile, a1[ti], c, q);
    __syncthreads();
    f32x4 zt[TN_MAXL], at[TN_MAXL];
    {
        f32x4 z = *(const f32x4*)(s_zred + ln * 4);
#pragma unroll
        for (int w = 1; w < NW; ++w) z += *(const f32x4*)(s_zred + w * 256 + ln * 4);
        z += bl[0];
        zt[0] = z;
        at[0] = L == 1 ? (SIG ? z : tn_act4(d.lastact, z)) : tn_hidden_act<SIG>(y0.act, z);
    }
    // ---- deeper tail layers (redundantly in every wave)
#pragma unroll
    for (int l = 1; l < L; ++l) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wfl[l][r], at[l - 1][r], acc, 0, 0, 0);
        acc += bl[l];
        zt[l] = acc;
        at[l] = l == L - 1 ? (SIG ? acc : tn_act4(d.lastact, acc)) : tn_hidden_act<SIG>(d.lay[l].act, acc);
    }
    // ---- loss + delta of the last layer (RL: lane (sample c, q) holds outputs 4q + r)
    f32x4 dl[TN_MAXL];
    float lsum = 0.f;
    {
        const f32x4 da = SIG ? (f32x4){1.f, 1.f, 1.f, 1.f} : tn_dact4(d.lastact, zt[L - 1], at[L - 1]);
        f32x4 dd;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = 4 * q + r < nout && rvalid;
            const float res = (at[L - 1][r] - tgt[r]) * lw[r];
            const float v = io.cot_mode ? tgt[r] * da[r] : 2.f * res * lw[r] * io.inv_div * da[r];
            lsum += (ok && !io.cot_mode) ? res * res : 0.f;
            dd[r] = ok ? v : 0.f;
        }
        dl[L - 1] = dd;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o, 64);
    const __amdgpu_buffer_rsrc_t rp = tn_rsrc(io.part_out, (int)(d.PS * 4));
    if (wv == 0 && ln == 0) tn_stw<AUX>(rp, 0, lsum);
    // ---- backward through the deeper layers (every wave; wave 0 stores the small gradients)
#pragma unroll
    for (int l = L - 1; l >= 1; --l) {
        const TailNLay y = d.lay[l];
        const f32x4 dc = tn_transpose(tile, dl[l], c, q);              // CL of delta_l: lane (n = c, q) holds delta[sample 4q + r][n]
        const f32x4 ac = tn_transpose(tile, at[l - 1], c, q);          // CL of A_{l-1}: lane (k = c, q) holds A[sample 4q + r][k]
        f32x4 gw = {0.f, 0.f, 0.f, 0.f}, dp = gw;
#pragma unroll
        for (int r = 0; r < 4; ++r) gw = __builtin_amdgcn_mfma_f32_16x16x4f32(dc[r], ac[r], gw, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) dp = __builtin_amdgcn_mfma_f32_16x16x4f32(wbl[l][r], dl[l][r], dp, 0, 0, 0);
        dl[l - 1] = dp * tn_hidden_dact<SIG>(d.lay[l - 1].act, zt[l - 1], at[l - 1]);     // (padding outputs: dp = 0)
        if (wv == 0) {
            // gw: lane (k = c, q) holds dW[n = 4q + r][k]
            if (c < y.K) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (okn[l][r]) tn_stw<AUX>(rp, (int)y.pW + (4 * q + r) + c * y.N, gw[r]);
            }
            float db = (dc[0] + dc[1]) + (dc[2] + dc[3]);
            db += __shfl_xor(db, 16, 64);
            db += __shfl_xor(db, 32, 64);
            if (q == 0 && c < y.N) tn_stw<AUX>(rp, (int)y.pb + c, db);
        }
    }
    // ---- first tail layer: dW tiles and delta_1 tiles split by wave
    {
        const f32x4 dc = tn_transpose(tile, dl[0], c, q);
        if (wv == 0) {
            float db = (dc[0] + dc[1]) + (dc[2] + dc[3]);
            db += __shfl_xor(db, 16, 64);
            db += __shfl_xor(db, 32, 64);
            if (q == 0 && c < y0.N) tn_stw<AUX>(rp, (int)y0.pb + c, db);
        }
        const int nb4 = nt >> 2;
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
            const int t = wv + NW * ti;
            if (t < nt) {                                           // wave-uniform
                f32x4 gw = {0.f, 0.f, 0.f, 0.f}, dp = gw;
#pragma unroll
                for (int r = 0; r < 4; ++r) gw = __builtin_amdgcn_mfma_f32_16x16x4f32(dc[r], a1c[ti][r], gw, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) dp = __builtin_amdgcn_mfma_f32_16x16x4f32(wb0[ti][r], dl[0][r], dp, 0, 0, 0);
                const int k = 16 * t + c;                           // gw: lane (k16 = c, q) holds dW[n = 4q + r][k]
                if (k < H1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (okn[0][r]) tn_stw<AUX>(rp, (int)y0.pW + (4 * q + r) + k * y0.N, gw[r]);
                }
                // delta_1 RL tile (columns >= H1 are 0: dp = 0 there, the weights read past the end)
                const f32x4 d1 = dp * tn_hidden_dact<SIG>(d.act0, z1[ti], a1[ti]);
                if (EP) {
                    // natural column order, one 16-byte write-through store per lane
                    const __amdgpu_buffer_rsrc_t rd = tn_rsrc(io.delta1, 0x7fffffff);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tn_u32x4, d1), rd, (row_g * H1p + 16 * t + 4 * q) * 4, 0, 16);
                } else {
                    // LDS image in isk_delta1_pos order (sample row c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int o = 16 * t + 4 * q + r;
                        const int pos = t < 4 * nb4 ? 64 * (t >> 2) + 4 * (4 * q + r) + (t & 3) : o;
                        s_d1[c * H1p + pos] = d1[r];
                    }
                }
            }
        }
    }
    if (!EP) {
        __syncthreads();
        const int tid = wv * 64 + ln;
        const int row = tid >> 4, nq = H1p >> 2;                    // 16 threads per sample row, float4 pieces
        if (row < 16) {
            float* drow = io.delta1 + (long long)(unit * 16 + row) * H1p;
            for (int p = tid & 15; p < nq; p += 16) *(f32x4*)(drow + 4 * p) = *(const f32x4*)(s_d1 + row * H1p + 4 * p);
        }
    }
    return lsum;
}

