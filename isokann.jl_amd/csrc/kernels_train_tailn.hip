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
#include "isk_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TN_WAVES 4
#define TN_TPW 3                    /* layer-1 tiles per wave: nt1 <= 12 */
#define TN_MAXL 3                   /* tail layers */
#define TN_TS 20                    /* row stride (floats) of a transpose tile: 16-byte aligned rows, spread over the banks */

struct TailNLay {
    int K, N, act;
    long long gW, gb;               // offsets in theta (Flux layout W[n + k*N])
    long long pW, pb;               // offsets in the per-unit partial vector
};
struct TailNDesc {
    int L, H1, H1p, nt, nout, act0, lastact;
    long long off_b0, PS;
    TailNLay lay[TN_MAXL];
};

__device__ __forceinline__ float tn_sigmoid(float z) {
    const float e = __expf(-fabsf(z));
    const float r = __builtin_amdgcn_rcpf(1.0f + e);               // (1 ulp; an IEEE division is ten instructions per element)
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

// buffer loads: one instruction per load with a 32-bit per-lane byte offset, and an offset past the end returns 0 -- which is
// how every padding element (output n >= N, input k >= K) is masked: its offset is -1
__device__ __forceinline__ float tn_ldw(__amdgpu_buffer_rsrc_t r, bool ok, int idx) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, ok ? idx * 4 : -1, 0, 0));
}
typedef unsigned tn_u32x4 __attribute__((ext_vector_type(4)));

template <int L, bool SIG>
__global__ __launch_bounds__(TN_WAVES * 64) void train_tailn_kernel(TailNDesc d, TrainBuffers b, const int* __restrict__ perm, int B,
                                                                   float inv_div) {
    ISK_LATENCY_CHAIN_PRIO();
    __shared__ __attribute__((aligned(16))) float s_zred[TN_WAVES][256];
    __shared__ __attribute__((aligned(16))) float s_tile[TN_WAVES][16 * TN_TS];
    __shared__ __attribute__((aligned(16))) float s_d1[16 * 16 * TN_TPW * TN_WAVES];      // delta1 image [16][H1p <= 192]
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63, c = ln & 15, q = ln >> 4;
    const int unit = blockIdx.x;
    const int H1 = d.H1, H1p = d.H1p, nt = d.nt, nout = d.nout;
    float* const tile = s_tile[wv];
    // ---- prologue: every load of the kernel is requested here (one memory round trip, two for the target rows)
    const int row_g = unit * 16 + c;
    const bool rvalid = row_g < B;
    const int prow = perm ? perm[rvalid ? row_g : B - 1] : (rvalid ? row_g : B - 1);
    const TailNLay y0 = d.lay[0];
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)b.a1, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb1 = __builtin_amdgcn_make_buffer_rsrc((void*)(b.theta + d.off_b0), 0, H1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw0 = __builtin_amdgcn_make_buffer_rsrc((void*)(b.theta + y0.gW), 0, y0.K * y0.N * 4, 0x00020000);
    f32x4 zp[TN_TPW][ISK_FWD1_MAXKS];
    f32x4 b1v[TN_TPW], wf0[TN_TPW], wb0[TN_TPW];
    const int zs4 = (int)(b.zstride * 4);
#pragma unroll
    for (int ti = 0; ti < TN_TPW; ++ti) {
        const int t = wv + TN_WAVES * ti;
        const int tt = t < nt ? t : nt - 1;                        // absent tiles re-read the last one (never used)
        const int zoff = (row_g * H1p + 16 * tt + 4 * q) * 4;
#pragma unroll
        for (int ks = 0; ks < ISK_FWD1_MAXKS; ++ks)
            zp[ti][ks] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, zoff, (ks < b.KS ? ks : 0) * zs4, 0));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * tt + 4 * q + r;                     // this lane's layer-1 output column
            b1v[ti][r] = tn_ldw(rb1, true, k);                     // (k >= H1 is past the end: 0)
            wf0[ti][r] = tn_ldw(rw0, c < y0.N, c + k * y0.N);                                  // W_2[n = c][k]            (k >= H1: past the end)
            wb0[ti][r] = tn_ldw(rw0, 4 * q + r < y0.N, 4 * q + r + (16 * tt + c) * y0.N);      // W_2[n = 4q + r][k = 16 t + c]
        }
    }
    f32x4 wfl[TN_MAXL], wbl[TN_MAXL], bl[TN_MAXL];                 // deeper layers: one tile each; bl also for layer 0
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const TailNLay y = d.lay[l];
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(b.theta + y.gW), 0, y.K * y.N * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(b.theta + y.gb), 0, y.N * 4, 0x00020000);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = 4 * q + r;
            bl[l][r] = tn_ldw(rb, true, n);
            if (l > 0) {
                wfl[l][r] = tn_ldw(rw, c < y.N && n < y.K, c + n * y.N);         // W[n = c][k = 4q + r]
                wbl[l][r] = tn_ldw(rw, n < y.N && c < y.K, n + c * y.N);         // W[n = 4q + r][k = c]
            }
        }
    }
    f32x4 tgt, lw;
    {
        const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)b.target, 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)b.lossw, 0, b.cot_mode ? 0 : nout * 4, 0x00020000);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 4 * q + r;
            tgt[r] = tn_ldw(rt, o < nout, prow * nout + o);
            lw[r] = tn_ldw(rl, true, o);
        }
    }
    const bool bt_owner = b.bt && unit == 0 && tid == 0;
    double bt0 = 0.0, bt1 = 0.0;
    if (bt_owner) { bt0 = b.bt[0]; bt1 = b.bt[1]; }
    bool okn[TN_MAXL][4];                                           // output 4q + r exists in tail layer l
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int r = 0; r < 4; ++r) okn[l][r] = 4 * q + r < d.lay[l].N;

    // ---- layer 1: z1 = b1 + sum of the K-slice partials, a1 = act(z1); RL images of this wave's tiles (columns >= H1: 0)
    f32x4 z1[TN_TPW], a1[TN_TPW], a1c[TN_TPW];
#pragma unroll
    for (int ti = 0; ti < TN_TPW; ++ti) {
        const int t = wv + TN_WAVES * ti;
        f32x4 z = b1v[ti];
#pragma unroll
        for (int ks = 0; ks < ISK_FWD1_MAXKS; ++ks)
            if (ks < b.KS) z += zp[ti][ks];
        // no masks: a padding column (>= H1) has z = 0 exactly (its layer-1 weights, its bias read past the end) and every
        // weight that multiplies its activation -- forward, delta and dW alike -- is a past-the-end load, i.e. 0
        z1[ti] = z;
        a1[ti] = tn_hidden_act<SIG>(d.act0, z);
        (void)t;
    }
    // ---- first tail layer: partial Z' over this wave's tiles, summed across the waves (fixed order: every wave gets the same bits)
    {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
        for (int ti = 0; ti < TN_TPW; ++ti) {
            if (wv + TN_WAVES * ti < nt) {                          // wave-uniform
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf0[ti][r], a1[ti][r], acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wf0[ti][r], a1[ti][r], acc0, 0, 0, 0);
                }
            }
        }
        *(f32x4*)(&s_zred[wv][ln * 4]) = acc0 + acc1;
    }
    // the CL images of a1 (for dW of the first tail layer) are taken while the other waves arrive at the barrier
#pragma unroll
    for (int ti = 0; ti < TN_TPW; ++ti) a1c[ti] = tn_transpose(tile, a1[ti], c, q);
    __syncthreads();
    f32x4 zt[TN_MAXL], at[TN_MAXL];
    {
        f32x4 z = *(const f32x4*)(&s_zred[0][ln * 4]);
#pragma unroll
        for (int w = 1; w < TN_WAVES; ++w) z += *(const f32x4*)(&s_zred[w][ln * 4]);
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
            const float v = b.cot_mode ? tgt[r] * da[r] : 2.f * res * lw[r] * inv_div * da[r];
            lsum += (ok && !b.cot_mode) ? res * res : 0.f;
            dd[r] = ok ? v : 0.f;
        }
        dl[L - 1] = dd;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o, 64);
    float* part_out = b.part + (long long)unit * d.PS;
    if (tid == 0) part_out[0] = lsum;
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
                    if (okn[l][r]) part_out[y.pW + (4 * q + r) + c * y.N] = gw[r];
            }
            float db = (dc[0] + dc[1]) + (dc[2] + dc[3]);
            db += __shfl_xor(db, 16, 64);
            db += __shfl_xor(db, 32, 64);
            if (q == 0 && c < y.N) part_out[y.pb + c] = db;
        }
    }
    // ---- first tail layer: dW tiles and delta_1 tiles split by wave
    {
        const f32x4 dc = tn_transpose(tile, dl[0], c, q);
        if (wv == 0) {
            float db = (dc[0] + dc[1]) + (dc[2] + dc[3]);
            db += __shfl_xor(db, 16, 64);
            db += __shfl_xor(db, 32, 64);
            if (q == 0 && c < y0.N) part_out[y0.pb + c] = db;
        }
        const int nb4 = nt >> 2;
#pragma unroll
        for (int ti = 0; ti < TN_TPW; ++ti) {
            const int t = wv + TN_WAVES * ti;
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
                        if (okn[0][r]) part_out[y0.pW + (4 * q + r) + k * y0.N] = gw[r];
                }
                // delta_1 RL tile -> LDS image in isk_delta1_pos order (sample row c); columns >= H1 are 0 (z1 = a1 = 0 there
                // and the weights read past the end)
                const f32x4 d1 = dp * tn_hidden_dact<SIG>(d.act0, z1[ti], a1[ti]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = 16 * t + 4 * q + r;
                    const int pos = t < 4 * nb4 ? 64 * (t >> 2) + 4 * (4 * q + r) + (t & 3) : o;
                    s_d1[c * H1p + pos] = o < H1 ? d1[r] : 0.f;
                }
            }
        }
    }
    __syncthreads();
    {
        const int row = tid >> 4, nq = H1p >> 2;                    // 16 threads per sample row, float4 pieces
        float* drow = b.delta1 + (long long)(unit * 16 + row) * H1p;
        for (int p = tid & 15; p < nq; p += 16) *(f32x4*)(drow + 4 * p) = *(const f32x4*)(s_d1 + row * H1p + 4 * p);
    }
    if (bt_owner) {
        b.bt[2] = bt0; b.bt[3] = bt1;
        b.bt[0] = bt0 * b.beta1; b.bt[1] = bt1 * b.beta2;
    }
}

// narrow tails only: 1..TN_MAXL tail layers, every one at most 16 outputs wide, first hidden width <= 16 * TN_TPW * TN_WAVES
bool isk_train_tail_narrow_ok(const TrainShape& s) {
    const int L = s.nlayers - 1;
    if (L < 1 || L > TN_MAXL) return false;
    if (isk_pad16(s.widths[1]) > 160) return false;
    for (int l = 1; l <= L; ++l)
        if (s.widths[l + 1] > 16 || s.widths[l + 1] < 1) return false;
    return true;
}

void isk_launch_train_tail_narrow(const TrainShape& s, const TrainBuffers& b, const int* perm, int B, float inv_div, hipStream_t st) {
    TailNDesc d = {};
    const int L = s.nlayers - 1;
    d.L = L; d.H1 = s.widths[1]; d.H1p = isk_pad16(s.widths[1]); d.nt = d.H1p / 16; d.nout = s.widths[s.nlayers];
    d.act0 = s.act; d.lastact = s.lastact;
    d.off_b0 = s.off_b[0];
    long long poff = 1 + s.widths[1];
    for (int i = 0; i < L; ++i) {
        const int l = i + 1;
        TailNLay& y = d.lay[i];
        y.K = s.widths[l]; y.N = s.widths[l + 1];
        y.act = (l == s.nlayers - 1) ? s.lastact : s.act;
        y.gW = s.off_W[l]; y.gb = s.off_b[l];
        y.pW = poff; y.pb = poff + (long long)y.N * y.K;
        poff = y.pb + y.N;
    }
    d.PS = poff;
    const int units = (B + 15) / 16;
    // SIG: the pairnet defaults (sigmoid hidden layers, identity last layer) with the activations inlined
    const bool sig = s.act == ISK_ACT_SIGMOID && s.lastact == ISK_ACT_IDENTITY;
#define TNCASE(L_)                                                                                                                  \
    case L_:                                                                                                                        \
        if (sig) hipLaunchKernelGGL((train_tailn_kernel<L_, true>), dim3(units), dim3(TN_WAVES * 64), 0, st, d, b, perm, B, inv_div);   \
        else hipLaunchKernelGGL((train_tailn_kernel<L_, false>), dim3(units), dim3(TN_WAVES * 64), 0, st, d, b, perm, B, inv_div);      \
        break;
    switch (L) { TNCASE(1) TNCASE(2) default: TNCASE(3) }
#undef TNCASE
}
