// Training-path kernels for gfx950 (MI355X): one optimizer step of train_batch!
// (src/iso.jl:179-194 of the reference) is the chain
//   train_fwd1  -> z1 = W1g * xhat[batch] + bias1                       (f32 MFMA, split-K over 4 waves)
//   train_tail  -> tail layers fwd, weighted loss, backward to delta1,  (VALU, 16 samples / workgroup)
//                  per-workgroup partial sums of the small gradients
//   train_grad1 -> G = delta1' * xhat[batch]  (f32 MFMA), LayerNorm-affine unfold:
//                  dW1 = g.*G + beta*D', dg = sum_o W1.*G, dbeta = W1'*D; reduces the partials
//   opt_step    -> WeightDecay + Adam | Nesterov on the flat parameter buffer (Optimisers.jl rules)
//   fold        -> refresh the fragment-linear weights (kernels_fwd.hip)
// The LayerNorm standardisation has no trainable parameters, so it is hoisted out of the loop
// (xhat is cached); its affine part (g, beta) is folded into layer 1 on the way in and unfolded
// on the way back, which removes the dL/d(a0) GEMM entirely: only G (one GEMM) is needed.
#include <cstdlib>
#include "isk_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define WAVE 64

__device__ __forceinline__ float wave_sum_t(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ float t_sigmoid(float z) {
    float e = __expf(-fabsf(z));
    float r = 1.0f / (1.0f + e);
    return z >= 0.f ? r : e * r;
}
__device__ __forceinline__ float t_act(int act, float z) {
    switch (act) {
        case ISK_ACT_SIGMOID: return t_sigmoid(z);
        case ISK_ACT_TANH: return tanhf(z);
        case ISK_ACT_RELU: return fmaxf(z, 0.f);
        case ISK_ACT_SWISH: return z * t_sigmoid(z);
        default: return z;
    }
}
__device__ __forceinline__ float t_dact(int act, float z, float a) {
    switch (act) {
        case ISK_ACT_SIGMOID: return a * (1.f - a);
        case ISK_ACT_TANH: return 1.f - a * a;
        case ISK_ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case ISK_ACT_SWISH: { float s = t_sigmoid(z); return s + a * (1.f - s); }
        default: return 1.f;
    }
}

// ------------------------------------------------------------------ train_fwd1 -------
// grid (units, nt1); workgroup = FW1_WAVES waves splitting the feature (K) axis of one 16x16
// output tile; every wave issues the loads of up to FW1_GROUP chunks before its MFMAs (the step
// is latency-bound: B rows only), partial tiles are combined through LDS.
#define FW1_WAVES 8
#define FW1_GROUP 7
__global__ __launch_bounds__(FW1_WAVES * 64) void train_fwd1_kernel(const float* __restrict__ X, int Fp, int nch,
                                                                   const float* __restrict__ W1t,
                                                                   const float* __restrict__ bias1, int nt1,
                                                                   const int* __restrict__ perm, int B,
                                                                   float* __restrict__ z1) {
    ISK_LATENCY_CHAIN_PRIO();
    __shared__ float red[FW1_WAVES][256];
    // blocks are dealt round-robin over the 8 XCDs (b % 8 shares an XCD): give every XCD whole units,
    // so a unit's gathered xhat rows are fetched into ONE L2 and re-read there by its nt1 tile blocks
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int unit = xcd + 8 * (slot / nt1), t = slot % nt1;
    if (unit * 16 >= B) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    int row = unit * 16 + li;
    if (row > B - 1) row = B - 1;
    const long long xr = perm ? perm[row] : row;
    const float* x = X + xr * (long long)Fp + lq * 4;
    const f32x4* Wp = (const f32x4*)W1t + lane;
    const float bcol = bias1[t * 16 + li];               // issued early, consumed by wave 0 at the end
    const int cq = (nch + FW1_WAVES - 1) / FW1_WAVES;
    const int c0 = wave * cq;
    int c1 = c0 + cq;
    if (c1 > nch) c1 = nch;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int c = c0; c < c1; c += FW1_GROUP) {
        f32x4 xa[FW1_GROUP], wa[FW1_GROUP];
#pragma unroll
        for (int i = 0; i < FW1_GROUP; ++i) {
            const int ci = (c + i < c1) ? c + i : c1 - 1;
            xa[i] = *(const f32x4*)(x + ci * 16);
            wa[i] = Wp[(long long)(ci * nt1 + t) * 64];
        }
#pragma unroll
        for (int i = 0; i < FW1_GROUP; ++i) {
            if (c + i < c1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (i & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i][j], wa[i][j], acc1, 0, 0, 0);
                    else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i][j], wa[i][j], acc0, 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][lane * 4 + r] = acc0[r] + acc1[r];
    __syncthreads();
    if (wave == 0) {
        const int H1p = nt1 * 16;
        const int col = t * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < FW1_WAVES; ++w) v += red[w][lane * 4 + r];
            z1[(long long)(unit * 16 + lq * 4 + r) * H1p + col] = v + bcol;
        }
    }
}

void isk_launch_train_fwd1(const TrainShape& s, const TrainBuffers& b, const int* perm, int B, hipStream_t st) {
    const int units = (B + 15) / 16;
    const int nt1 = isk_pad16(s.widths[1]) / 16;
    const int upx = (units + 7) / 8;                      // units per XCD group
    hipLaunchKernelGGL(train_fwd1_kernel, dim3(8 * upx * nt1), dim3(FW1_WAVES * 64), 0, st, b.xhat_xs, b.Fp, b.Fp / 16, b.W1t,
                       b.bias1, nt1, perm, B, b.a1);
}

// generic row GEMM on the same kernel: out[n][16*t + j] = sum_k X[row(n)][k] * Wt(k, t) + bias, any layer
void isk_launch_gemm_rows(const float* X, int ldx, const float* Wt, const float* bias, int nt, const int* perm, int B, float* out,
                          hipStream_t st) {
    const int units = (B + 15) / 16;
    const int upx = (units + 7) / 8;
    hipLaunchKernelGGL(train_fwd1_kernel, dim3(8 * upx * nt), dim3(FW1_WAVES * 64), 0, st, X, ldx, ldx / 16, Wt, bias, nt, perm, B, out);
}

// ------------------------------------------------------------------ train_tail -------
// partial vector layout (per workgroup):
//   [0] loss, [1 .. 1+H1) unused (D = sum_s delta1[s][.] is rebuilt inside train_grad1), then for
//   l = 1..L-1: dW_l (Flux order), db_l
__host__ __device__ static inline long long part_size(const TrainShape& s) {
    long long n = 1 + s.widths[1];
    for (int l = 1; l < s.nlayers; ++l) n += (long long)s.widths[l + 1] * s.widths[l] + s.widths[l + 1];
    return n;
}

int isk_train_tail_nblocks(int B) { return (B + 15) / 16; }
long long isk_train_part_floats(const TrainShape& s, int maxB) { return part_size(s) * isk_train_tail_nblocks(maxB); }

// LDS carve of the tail kernel: [A_0 | Z_0 | A_1 | Z_1 | ... | W_1 | W_2 | ...]; A_l, Z_l are 16 x widths[l+1]
__host__ __device__ static inline int tail_off_act(const TrainShape& s, int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += 32 * s.widths[i + 1];
    return o;
}
__host__ __device__ static inline int tail_off_w(const TrainShape& s, int l) {
    int o = tail_off_act(s, s.nlayers);
    for (int i = 1; i < l; ++i) o += s.widths[i + 1] * s.widths[i];
    return o;
}

#define TAIL_THREADS 1024
__global__ __launch_bounds__(TAIL_THREADS) void train_tail_kernel(TrainShape s, TrainBuffers b, const int* __restrict__ perm,
                                                                  int B, float inv_div) {
    ISK_LATENCY_CHAIN_PRIO();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float lred[TAIL_THREADS / 64];
    constexpr int NTH = TAIL_THREADS;
    const int L = s.nlayers;
    const int tid = threadIdx.x;
    const int unit = blockIdx.x;
    const int H1 = s.widths[1];
    const int H1p = (H1 + 15) & ~15;
    // stage the tail weights (Flux layout W[o + k*N]) in LDS; 16 waves keep the loads in flight
    for (int l = 1; l < L; ++l) {
        float* Wl = sm + tail_off_w(s, l);
        const float* Wg = b.theta + s.off_W[l];
        const int n = s.widths[l + 1] * s.widths[l];
        for (int e0 = 0; e0 < n; e0 += NTH * 2) {
            float v[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { const int e = e0 + u * NTH + tid; v[u] = (e < n) ? Wg[e] : 0.f; }
#pragma unroll
            for (int u = 0; u < 2; ++u) { const int e = e0 + u * NTH + tid; if (e < n) Wl[e] = v[u]; }
        }
    }
    {
        float* A0 = sm;
        float* Z0 = sm + 16 * H1;
        const int act0 = (L == 1) ? s.lastact : s.act;
        for (int e0 = 0; e0 < 16 * H1; e0 += NTH * 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * NTH + tid;
                const int sidx = e / H1, o = e - sidx * H1;
                v[u] = (e < 16 * H1) ? b.a1[(long long)(unit * 16 + sidx) * H1p + o] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * NTH + tid;
                if (e < 16 * H1) { Z0[e] = v[u]; A0[e] = t_act(act0, v[u]); }
            }
        }
    }
    __syncthreads();
    for (int l = 1; l < L; ++l) {
        const int K = s.widths[l], N = s.widths[l + 1];
        const float* Ap = sm + tail_off_act(s, l - 1);
        float* Al = sm + tail_off_act(s, l);
        float* Zl = Al + 16 * N;
        const float* Wl = sm + tail_off_w(s, l);
        const float* bias = b.theta + s.off_b[l];
        const int actl = (l == L - 1) ? s.lastact : s.act;
        // each (sample, output) dot product is split over 4 adjacent lanes along k
        for (int p4 = tid; p4 < 16 * N * 4; p4 += NTH) {
            const int p = p4 >> 2, ks = p4 & 3;
            const int sidx = p / N, o = p - sidx * N;
            const float* arow = Ap + sidx * K;
            float z0 = 0.f, z1 = 0.f;
            int k = ks;
            for (; k + 4 < K; k += 8) { z0 += Wl[o + k * N] * arow[k]; z1 += Wl[o + (k + 4) * N] * arow[k + 4]; }
            if (k < K) z0 += Wl[o + k * N] * arow[k];
            float z = z0 + z1;
            z += __shfl_xor(z, 1, WAVE);
            z += __shfl_xor(z, 2, WAVE);
            if (ks == 0) {
                z += bias[o];
                Zl[p] = z;
                Al[p] = t_act(actl, z);
            }
        }
        __syncthreads();
    }
    // loss + delta of the last layer (overwrites Z_{L-1})
    const int nout = s.widths[L];
    float lsum = 0.f;
    {
        float* AL = sm + tail_off_act(s, L - 1);
        float* ZL = AL + 16 * nout;
        for (int p = tid; p < 16 * nout; p += NTH) {
            const int sidx = p / nout, o = p - sidx * nout;
            const int row = unit * 16 + sidx;
            const bool valid = row < B;
            const int rowc = valid ? row : B - 1;
            const long long prow = perm ? perm[rowc] : rowc;
            const float out = AL[p];
            float d = 0.f;
            if (b.cot_mode) {                              // pullback: delta_L = cotangent .* act'(z_L)
                if (valid) d = b.target[prow * nout + o] * t_dact(s.lastact, ZL[p], out);
            } else {
                const float w = b.lossw[o];
                const float r = (out - b.target[prow * nout + o]) * w;
                if (valid) {
                    lsum += r * r;
                    d = 2.f * r * w * inv_div * t_dact(s.lastact, ZL[p], out);
                }
            }
            ZL[p] = d;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o, WAVE);
    if ((tid & 63) == 0) lred[tid >> 6] = lsum;
    __syncthreads();
    float* part = b.part + (long long)unit * part_size(s);
    if (tid == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NTH / 64; ++w) t += lred[w];
        part[0] = t;
    }
    // backward through the tail
    for (int l = L - 1; l >= 1; --l) {
        const int K = s.widths[l], N = s.widths[l + 1];
        float* Ap = sm + tail_off_act(s, l - 1);
        float* Zp = Ap + 16 * K;
        const float* Zl = sm + tail_off_act(s, l) + 16 * N;
        const float* Wl = sm + tail_off_w(s, l);
        long long poff = 1 + H1;
        for (int i = 1; i < l; ++i) poff += (long long)s.widths[i + 1] * s.widths[i] + s.widths[i + 1];
        float* pW = part + poff;
        float* pb = pW + (long long)N * K;
        for (int e = tid; e < N * K; e += NTH) {
            const int k = e / N, o = e - k * N;
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc += Zl[q * N + o] * Ap[q * K + k];
            pW[e] = acc;
        }
        for (int o = tid; o < N; o += NTH) {
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc += Zl[q * N + o];
            pb[o] = acc;
        }
        for (int p = tid; p < 16 * K; p += NTH) {
            const int sidx = p / K, k = p - sidx * K;
            const float* drow = Zl + sidx * N;
            float a0 = 0.f, a1 = 0.f;
            int o = 0;
            for (; o + 1 < N; o += 2) { a0 += Wl[o + k * N] * drow[o]; a1 += Wl[o + 1 + k * N] * drow[o + 1]; }
            if (o < N) a0 += Wl[o + k * N] * drow[o];
            Zp[p] = (a0 + a1) * t_dact(s.act, Zp[p], Ap[p]);      // read and written by this thread only
        }
        __syncthreads();
    }
    // delta1 -> global (zero padded)
    {
        const float* Z0 = sm + 16 * H1;
        for (int e = tid; e < 16 * H1p; e += NTH) {
            const int sidx = e / H1p, o = e - sidx * H1p;
            b.delta1[(long long)(unit * 16 + sidx) * H1p + o] = (o < H1) ? Z0[sidx * H1 + o] : 0.f;
        }
    }
}

static size_t tail_lds_bytes(const TrainShape& s) { return (size_t)tail_off_w(s, s.nlayers) * sizeof(float); }

void isk_launch_train_tail(const TrainShape& s, const TrainBuffers& b, const int* perm, int B, float inv_div,
                           hipStream_t st) {
    const int units = (B + 15) / 16;
    const size_t lds = tail_lds_bytes(s);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)train_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(train_tail_kernel, dim3(units), dim3(TAIL_THREADS), lds, st, s, b, perm, B, inv_div);
}

// ------------------------------------------------------------------ train_grad1 ------
// grid = Fp/16 feature tiles; workgroup = NW waves that SPLIT THE BATCH ROWS (the step is
// latency-bound, so every SIMD gets 4 waves of independent gathers); each wave accumulates all
// NT output tiles:  Gt[f][o] = sum_n xhat[n][f] * delta1[n][o]  (f32 MFMA), the NW partial tiles are
// combined through LDS, then waves 0..NT-1 run the LayerNorm-affine unfold of their tile.
// Output-column mapping: tiles are loaded four at a time with one dwordx4 per row, so tile t,
// lane column li holds o = 64*(t/4) + 4*li + t%4 for the first 4*(NT/4) tiles, 16*t + li after.
__device__ __forceinline__ long long part_to_grad(const TrainShape& s, long long e) {
    if (e == 0) return s.nparams;                       // loss slot
    e -= 1;
    if (e < s.widths[1]) return s.off_b[0] + e;          // D == db1
    e -= s.widths[1];
    for (int l = 1; l < s.nlayers; ++l) {
        const long long nw = (long long)s.widths[l + 1] * s.widths[l];
        if (e < nw) return s.off_W[l] + e;
        e -= nw;
        if (e < s.widths[l + 1]) return s.off_b[l] + e;
        e -= s.widths[l + 1];
    }
    return -1;
}

template <int NT, int NW>
__global__ __launch_bounds__(NW * 64) void train_grad1_kernel(TrainShape s, TrainBuffers b, const int* __restrict__ perm,
                                                             int B, int nparts) {
    ISK_LATENCY_CHAIN_PRIO();
    constexpr int NB4 = NT / 4, NR = NT % 4;
    constexpr int CB = 4;                          // chunks per batch: their row indices are fetched together
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* red = sm;                               // [NW][NT][256]
    float* Dred = sm + NW * NT * 256;              // [NW][NT][16]
    float* Dsh = Dred + NW * NT * 16;              // [NT*16] in tile order
    const int H1 = s.widths[1], F = s.widths[0];
    const int H1p = NT * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int ft = blockIdx.x;
    const int slab = blockIdx.y, nslab = gridDim.y;     // batch rows are split over `nslab` workgroups per feature tile;
    float* const gout = b.grad + (long long)slab * (s.nparams + 2);   // each writes its (linear) share to its own gradient slab
    const long long PS = part_size(s);
    const bool ln = s.layernorm != 0;
    // small-gradient partial vectors: one element per 16-lane group, lanes split the partials (slab 0 only)
    if (slab == 0) {
        const long long ngroups = (long long)gridDim.x * (NW * 4);
        const long long grp = (long long)blockIdx.x * (NW * 4) + (tid >> 4);
        for (long long e = grp; e < PS; e += ngroups) {
            if (e >= 1 && e < 1 + H1) continue;     // D is rebuilt below
            float acc = 0.f;
            for (int w0 = li; w0 < nparts; w0 += 64) {       // 4 independent loads per lane in flight
                float pv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int w = w0 + 16 * u; pv[u] = (w < nparts) ? b.part[(long long)w * PS + e] : 0.f; }
                acc += (pv[0] + pv[1]) + (pv[2] + pv[3]);
            }
#pragma unroll
            for (int sft = 8; sft > 0; sft >>= 1) acc += __shfl_xor(acc, sft, WAVE);
            const long long gi = part_to_grad(s, e);
            if (li == 0 && gi >= 0) gout[gi] = acc;
        }
    }
    f32x4 acc[NT];
    float dsum[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; dsum[t] = 0.f; }
    const int nchunks = (B + 15) / 16;
    const float* Xc = b.xhat_xs + ft * 16 + li;
    for (int c0 = wave + NW * slab; c0 < nchunks; c0 += NW * CB * nslab) {
        long long rr[CB][4];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const int chunk = c0 + cb * NW * nslab;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = chunk * 16 + 4 * lq + j;
                rr[cb][j] = (r < B) ? (perm ? perm[r] : r) : 0;      // rows >= B carry delta == 0
            }
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const int chunk = c0 + cb * NW * nslab;
            if (chunk < nchunks) {                                    // wave-uniform
                const int r0 = chunk * 16 + 4 * lq;
                float av[4];
                f32x4 bq[4][NB4 > 0 ? NB4 : 1];
                float br[4][NR > 0 ? NR : 1];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    av[j] = Xc[rr[cb][j] * (long long)b.Fp];
                    const float* drow = b.delta1 + (long long)(r0 + j) * H1p;
#pragma unroll
                    for (int q = 0; q < NB4; ++q) bq[j][q] = *(const f32x4*)(drow + 64 * q + 4 * li);
#pragma unroll
                    for (int u = 0; u < NR; ++u) br[j][u] = drow[64 * NB4 + 16 * u + li];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int q = 0; q < NB4; ++q)
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            acc[4 * q + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bq[j][q][u], acc[4 * q + u], 0, 0, 0);
                            dsum[4 * q + u] += bq[j][q][u];
                        }
#pragma unroll
                    for (int u = 0; u < NR; ++u) {
                        acc[4 * NB4 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], br[j][u], acc[4 * NB4 + u], 0, 0, 0);
                        dsum[4 * NB4 + u] += br[j][u];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        *(f32x4*)(red + ((wave * NT + t) * 64 + lane) * 4) = acc[t];
        float d = dsum[t];
        d += __shfl_xor(d, 16, WAVE);
        d += __shfl_xor(d, 32, WAVE);
        if (lq == 0) Dred[(wave * NT + t) * 16 + li] = d;
    }
    __syncthreads();
    if (tid < NT * 16) {                                               // D in tile order: Dsh[t*16 + li]
        float d = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) d += Dred[w * NT * 16 + tid];
        Dsh[tid] = d;
        const int t = tid >> 4, l2 = tid & 15;
        const int o = (t < 4 * NB4) ? 64 * (t >> 2) + 4 * l2 + (t & 3) : 16 * t + l2;
        if (ft == 0 && o < H1) gout[s.off_b[0] + o] = d;                // db1 == D (this slab's rows)
    }
    __syncthreads();
    // Epilogue, coalesced: wave w owns feature row f = 16*ft + w (NW >= 16 waves cover the tile; with
    // NW == 8 each wave takes two rows) and its lanes walk the outputs o, so the W1 reads and the dW1
    // writes are contiguous in Flux's (out, in) layout.  G[f][o] is re-assembled from the NW partial
    // accumulator tiles in LDS: C layout row = 4*lq + r is the feature, col = li the output slot.
    const float* W1 = b.theta + s.off_W[0];
    const float* g = ln ? b.theta + s.off_lns : nullptr;
    const float* beta = ln ? b.theta + s.off_lnb : nullptr;
    float* gW1 = gout + s.off_W[0];
    for (int fl = wave; fl < 16; fl += NW) {
        const int f = ft * 16 + fl;
        if (f >= F) continue;                                   // wave-uniform
        const int flq = fl >> 2, fr = fl & 3;
        const float gf = ln ? g[f] : 1.f, bf = ln ? beta[f] : 0.f;
        float pg = 0.f, pb = 0.f;
        for (int o = lane; o < H1; o += WAVE) {
            int t, l2;
            if (o < 64 * NB4) { t = 4 * (o >> 6) + (o & 3); l2 = (o & 63) >> 2; }
            else { t = 4 * NB4 + ((o - 64 * NB4) >> 4); l2 = o & 15; }
            const int slot = ((t * 64) + flq * 16 + l2) * 4 + fr;
            float G = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) G += red[w * NT * 256 + slot];
            if (ln) {
                const float w1 = W1[o + (long long)f * H1];
                const float Do = Dsh[t * 16 + l2];
                gW1[o + (long long)f * H1] = gf * G + bf * Do;
                pg += w1 * G;
                pb += w1 * Do;
            } else {
                gW1[o + (long long)f * H1] = G;
            }
        }
        if (ln) {
            pg = wave_sum_t(pg);
            pb = wave_sum_t(pb);
            if (lane == 0) { gout[s.off_lns + f] = pg; gout[s.off_lnb + f] = pb; }
        }
    }
}

template <int NT, int NW>
static void launch_grad1(const TrainShape& s, const TrainBuffers& b, const int* perm, int B, int nslab, hipStream_t st) {
    const int nparts = (B + 15) / 16;
    const size_t lds = (size_t)(NW * NT * 256 + NW * NT * 16 + NT * 16) * sizeof(float);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)train_grad1_kernel<NT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((train_grad1_kernel<NT, NW>), dim3(b.Fp / 16, nslab), dim3(NW * 64), lds, st, s, b, perm, B, nparts);
}

void isk_launch_train_grad1(const TrainShape& s, const TrainBuffers& b, const int* perm, int B, int nslab_flags, hipStream_t st) {
    const int nt1 = isk_pad16(s.widths[1]) / 16;
    const int nslab = nslab_flags & 0xff;
    static const int nw8_env = [] { const char* v = getenv("ISK_GRAD1_NW8"); return v ? atoi(v) : -1; }();
    const int nw8 = nw8_env >= 0 ? nw8_env : ((nslab_flags & 0x100) != 0);
#define G1CASE(NT_) case NT_: if (nw8) launch_grad1<NT_, 8>(s, b, perm, B, nslab, st); else launch_grad1<NT_, 16>(s, b, perm, B, nslab, st); break;
    switch (nt1) {
        G1CASE(1) G1CASE(2) G1CASE(3) G1CASE(4) G1CASE(5) G1CASE(6) G1CASE(7) G1CASE(8) G1CASE(9)
        default: launch_grad1<10, 8>(s, b, perm, B, nslab, st); break;
    }
#undef G1CASE
}

// ------------------------------------------------------------------ opt_step ---------
// Optimisers.jl 0.4.7: OptimiserChain(WeightDecay(reg), Adam | Nesterov), one thread per parameter.
// The batch loss sits in grad[n]; a non-finite loss (or an earlier one in this epoch) skips the
// update, as the reference throws before update! (src/iso.jl:186-189).
__global__ __launch_bounds__(256) void opt_step_kernel(OptArgs a, double* loss_acc) {
    ISK_LATENCY_CHAIN_PRIO();
    const float bc1 = a.bt ? (float)(1.0 - a.bt[0]) : a.bc1, bc2 = a.bt ? (float)(1.0 - a.bt[1]) : a.bc2;
    const float lstep = a.grad[a.n];
    const bool bad = (loss_acc[1] != 0.0) || !isfinite(lstep);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (bad) loss_acc[1] = 1.0;
        else loss_acc[0] += (double)lstep;
    }
    if (bad) return;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256) {
        const float th = a.theta[i];
        float gs = a.grad[i];
        if (i < a.slab_prefix)                           // [LN | W1 | b1] gradients arrive split over row slabs
            for (int sl = 1; sl < a.nslab; ++sl) gs += a.grad[(long long)sl * (a.n + 2) + i];
        const float g = gs * a.gscale + a.reg * th;
        float d;
        if (a.kind == ISK_OPT_ADAM) {
            const float m = a.beta1 * a.m[i] + (1.f - a.beta1) * g;
            const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
            a.m[i] = m;
            a.v[i] = v;
            d = m / bc1 / (sqrtf(v / bc2) + a.eps) * a.lr;
        } else {
            const float vel = a.m[i];
            d = -a.rho * a.rho * vel + (1.f + a.rho) * a.lr * g;
            a.m[i] = a.rho * vel - a.lr * g;
        }
        a.theta[i] = th - d;
    }
}

void isk_launch_opt_step(const OptArgs& a, hipStream_t st) {
    long long b = (a.n + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(opt_step_kernel, dim3((unsigned)b), dim3(256), 0, st, a, (double*)a.loss_acc);
}

// ------------------------------------------------------------------ coordinate pullback ---
// one workgroup per sample column:
//   dxhat[f] = g[f] * sum_o delta1[o] * W1[o][f]                       (layer 1 + LayerNorm scale)
//   dfeat[f] = rstd * (dxhat[f] - mean(dxhat) - xhat[f] * mean(dxhat .* xhat) * c)   (standardisation; c = 1, or
//              (std+eps)/std for the (x-mu)/(std+eps) variant)
//   grad[a] += dfeat[f] * (x_a - x_b) / d_f,  grad[b] -= ...           (pairwise distances; LDS float atomics)
__global__ __launch_bounds__(256) void vjp_back_kernel(VjpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[2][4];
    float* dl = sm;                        // [H1p]  delta1 of this sample
    float* dx = sm + a.H1p;                // [Fp]   dxhat, then dfeat
    float* gc = dx + a.Fp;                 // [D]    coordinate gradient accumulator (pdist) 
    const long long n = blockIdx.x;
    const int tid = threadIdx.x;
    const bool ln = a.off_lns >= 0;
    for (int o = tid; o < a.H1p; o += 256) dl[o] = a.delta1[n * a.H1p + o];
    if (a.pairs)
        for (int i = tid; i < a.D; i += 256) gc[i] = 0.f;
    __syncthreads();
    const float* W1 = a.theta + a.off_W0;
    const float* g = ln ? a.theta + a.off_lns : nullptr;
    const float* xh = a.xhat + n * a.Fp;
    float s1 = 0.f, s2 = 0.f;
    for (int f = tid; f < a.F; f += 256) {
        const float* w = W1 + (long long)f * a.H1;
        float acc0 = 0.f, acc1 = 0.f;
        int o = 0;
        for (; o + 1 < a.H1; o += 2) { acc0 += dl[o] * w[o]; acc1 += dl[o + 1] * w[o + 1]; }
        if (o < a.H1) acc0 += dl[o] * w[o];
        float v = acc0 + acc1;
        if (ln) v *= g[f];
        dx[f] = v;
        s1 += v;
        s2 += v * xh[f];
    }
    if (ln) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, WAVE); s2 += __shfl_xor(s2, o, WAVE); }
        if ((tid & 63) == 0) { red[0][tid >> 6] = s1; red[1][tid >> 6] = s2; }
        __syncthreads();
        const float m1 = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)a.F;
        float m2 = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)a.F;
        const float rstd = a.stats[2 * n + 1];
        if (a.ln_mode == ISK_LN_STD_EPS) {
            // xhat = (x-mu)/(std+eps): d(std+eps)/dx_j = (x_j-mu)/(F*std)
            float var = 0.f;   // recover std from xhat: mean(xhat^2) = var/(std+eps)^2
            float q = 0.f;
            for (int f = tid; f < a.F; f += 256) q += xh[f] * xh[f];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, WAVE);
            __syncthreads();
            if ((tid & 63) == 0) red[0][tid >> 6] = q;
            __syncthreads();
            var = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)a.F;   // = std^2 * rstd^2
            const float stdv = sqrtf(var) / rstd;
            m2 *= (stdv > 0.f) ? (1.0f / rstd) / stdv : 0.f;
        }
        for (int f = tid; f < a.F; f += 256) dx[f] = rstd * (dx[f] - m1 - xh[f] * m2);
    }
    __syncthreads();
    if (!a.pairs) {                        // identity featurizer: the input IS the feature vector
        float* out = a.grad + n * (long long)a.D;
        for (int f = tid; f < a.F; f += 256) out[f] = dx[f];
        return;
    }
    const float* c = a.coords + n * (long long)a.D;
    const float* raw = a.raw + n * a.Fp;
    for (int f = tid; f < a.F; f += 256) {
        const int2 p = a.pairs[f];
        const float d = raw[f];
        const float sc = d > 0.f ? dx[f] / d : 0.f;        // d|x_a - x_b|/dx is undefined at 0: subgradient 0, as sqrt'(0)*0
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = sc * (c[3 * p.x + k] - c[3 * p.y + k]);
            atomicAdd(&gc[3 * p.x + k], v);
            atomicAdd(&gc[3 * p.y + k], -v);
        }
    }
    __syncthreads();
    float* out = a.grad + n * (long long)a.D;
    for (int i = tid; i < a.D; i += 256) out[i] = gc[i];
}

void isk_launch_vjp_back(const VjpArgs& a, hipStream_t st) {
    if (a.S <= 0) return;
    const size_t lds = (size_t)(a.H1p + a.Fp + (a.pairs ? a.D : 0)) * sizeof(float);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)vjp_back_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(vjp_back_kernel, dim3((unsigned)a.S), dim3(256), lds, st, a);
}
