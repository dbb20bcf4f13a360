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
#include "isk_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define WAVE 64

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
// grid (units, nt1); workgroup = 4 waves splitting the feature (K) axis; one 16x16 output tile.
__global__ __launch_bounds__(256) void train_fwd1_kernel(const float* __restrict__ X, int Fp, int nch,
                                                         const float* __restrict__ W1t, const float* __restrict__ bias1,
                                                         int nt1, const int* __restrict__ perm, int B,
                                                         float* __restrict__ z1) {
    __shared__ float red[4][256];
    const int unit = blockIdx.x, t = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    int row = unit * 16 + li;
    if (row > B - 1) row = B - 1;
    const long long xr = perm ? perm[row] : row;
    const float* x = X + xr * (long long)Fp + lq * 4;
    const f32x4* Wp = (const f32x4*)W1t + lane;
    const int cq = (nch + 3) / 4;
    const int c0 = wave * cq;
    int c1 = c0 + cq;
    if (c1 > nch) c1 = nch;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int c = c0;
#pragma unroll 2
    for (; c + 1 < c1; c += 2) {
        const f32x4 xa = *(const f32x4*)(x + c * 16);
        const f32x4 wa = Wp[(long long)(c * nt1 + t) * 64];
        const f32x4 xb = *(const f32x4*)(x + (c + 1) * 16);
        const f32x4 wb = Wp[(long long)((c + 1) * nt1 + t) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[j], wa[j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[j], wb[j], acc1, 0, 0, 0);
        }
    }
    if (c < c1) {
        const f32x4 xa = *(const f32x4*)(x + c * 16);
        const f32x4 wa = Wp[(long long)(c * nt1 + t) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[j], wa[j], acc0, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][lane * 4 + r] = acc0[r] + acc1[r];
    __syncthreads();
    if (wave == 0) {
        const int H1p = nt1 * 16;
        const int col = t * 16 + li;
        const float bcol = bias1[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = red[0][lane * 4 + r] + red[1][lane * 4 + r] + red[2][lane * 4 + r] + red[3][lane * 4 + r];
            z1[(long long)(unit * 16 + lq * 4 + r) * H1p + col] = v + bcol;
        }
    }
}

void isk_launch_train_fwd1(const TrainShape& s, const TrainBuffers& b, const int* perm, int B, hipStream_t st) {
    const int units = (B + 15) / 16;
    const int nt1 = isk_pad16(s.widths[1]) / 16;
    hipLaunchKernelGGL(train_fwd1_kernel, dim3(units, nt1), dim3(256), 0, st, b.xhat_xs, b.Fp, b.Fp / 16, b.W1t, b.bias1,
                       nt1, perm, B, b.a1);
}

// ------------------------------------------------------------------ train_tail -------
// partial vector layout (per workgroup):
//   [0] loss, [1 .. 1+H1) D = sum_s delta1[s][.], then for l = 1..L-1: dW_l (Flux order), db_l
__host__ __device__ static inline long long part_size(const TrainShape& s) {
    long long n = 1 + s.widths[1];
    for (int l = 1; l < s.nlayers; ++l) n += (long long)s.widths[l + 1] * s.widths[l] + s.widths[l + 1];
    return n;
}

int isk_train_tail_nblocks(int B) { return (B + 15) / 16; }
long long isk_train_part_floats(const TrainShape& s, int maxB) { return part_size(s) * isk_train_tail_nblocks(maxB); }

__global__ __launch_bounds__(256) void train_tail_kernel(TrainShape s, TrainBuffers b, const int* __restrict__ perm, int B,
                                                         float inv_div) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float lred[4];
    const int L = s.nlayers;
    const int tid = threadIdx.x, sidx = tid >> 4, t = tid & 15;
    const int unit = blockIdx.x;
    const int row = unit * 16 + sidx;
    const bool valid = row < B;
    const int rowc = valid ? row : B - 1;
    const long long prow = perm ? perm[rowc] : rowc;
    // LDS carve: A[l], Z[l] for l = 0..L-1, each 16 x widths[l+1]
    float* A[ISK_MAX_LAYERS];
    float* Z[ISK_MAX_LAYERS];
    {
        float* p = sm;
        for (int l = 0; l < L; ++l) { A[l] = p; p += 16 * s.widths[l + 1]; Z[l] = p; p += 16 * s.widths[l + 1]; }
    }
    const int H1 = s.widths[1];
    const int H1p = (H1 + 15) & ~15;
    const int act0 = (L == 1) ? s.lastact : s.act;
    for (int o = t; o < H1; o += 16) {
        const float z = b.a1[(long long)row * H1p + o];
        Z[0][sidx * H1 + o] = z;
        A[0][sidx * H1 + o] = t_act(act0, z);
    }
    __syncthreads();
    for (int l = 1; l < L; ++l) {
        const int K = s.widths[l], N = s.widths[l + 1];
        const float* W = b.theta + s.off_W[l];
        const float* bias = b.theta + s.off_b[l];
        const int actl = (l == L - 1) ? s.lastact : s.act;
        for (int o = t; o < N; o += 16) {
            float z = bias[o];
            for (int k = 0; k < K; ++k) z += W[o + (long long)k * N] * A[l - 1][sidx * K + k];
            Z[l][sidx * N + o] = z;
            A[l][sidx * N + o] = t_act(actl, z);
        }
        __syncthreads();
    }
    // loss + delta of the last layer
    const int nout = s.widths[L];
    const int actL = s.lastact;
    float lsum = 0.f;
    for (int o = t; o < nout; o += 16) {
        const float w = b.lossw[o];
        const float out = A[L - 1][sidx * nout + o];
        const float r = (out - b.target[prow * nout + o]) * w;
        float d = 0.f;
        if (valid) {
            lsum += r * r;
            d = 2.f * r * w * inv_div * t_dact(actL, Z[L - 1][sidx * nout + o], out);
        }
        Z[L - 1][sidx * nout + o] = d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o, WAVE);
    if ((tid & 63) == 0) lred[tid >> 6] = lsum;
    __syncthreads();
    float* part = b.part + (long long)unit * part_size(s);
    if (tid == 0) part[0] = lred[0] + lred[1] + lred[2] + lred[3];
    // backward through the tail
    long long poff = 1 + H1;
    // offsets of each tail layer inside the partial vector
    long long poffs[ISK_MAX_LAYERS];
    for (int l = 1; l < L; ++l) { poffs[l] = poff; poff += (long long)s.widths[l + 1] * s.widths[l] + s.widths[l + 1]; }
    for (int l = L - 1; l >= 1; --l) {
        const int K = s.widths[l], N = s.widths[l + 1];
        const float* W = b.theta + s.off_W[l];
        float* pW = part + poffs[l];
        float* pb = pW + (long long)N * K;
        for (int e = tid; e < N * K; e += 256) {
            const int o = e % N, k = e / N;
            float acc = 0.f;
#pragma unroll 4
            for (int q = 0; q < 16; ++q) acc += Z[l][q * N + o] * A[l - 1][q * K + k];
            pW[e] = acc;
        }
        for (int o = tid; o < N; o += 256) {
            float acc = 0.f;
            for (int q = 0; q < 16; ++q) acc += Z[l][q * N + o];
            pb[o] = acc;
        }
        const int actp = s.act;      // layer l-1 is a hidden layer
        for (int k = t; k < K; k += 16) {
            float acc = 0.f;
            for (int o = 0; o < N; ++o) acc += W[o + (long long)k * N] * Z[l][sidx * N + o];
            const float zk = Z[l - 1][sidx * K + k];
            const float ak = A[l - 1][sidx * K + k];
            // Z[l-1][sidx][k] is read and written by this thread only
            Z[l - 1][sidx * K + k] = acc * t_dact(actp, zk, ak);
        }
        __syncthreads();
    }
    // delta1 -> global (zero padded), D partial
    for (int o = t; o < H1p; o += 16)
        b.delta1[(long long)row * H1p + o] = (o < H1) ? Z[0][sidx * H1 + o] : 0.f;
    for (int o = tid; o < H1; o += 256) {
        float acc = 0.f;
        for (int q = 0; q < 16; ++q) acc += Z[0][q * H1 + o];
        part[1 + o] = acc;
    }
}

static size_t tail_lds_bytes(const TrainShape& s) {
    size_t n = 0;
    for (int l = 0; l < s.nlayers; ++l) n += (size_t)2 * 16 * s.widths[l + 1];
    return n * sizeof(float);
}

void isk_launch_train_tail(const TrainShape& s, const TrainBuffers& b, const int* perm, int B, float inv_div,
                           hipStream_t st) {
    const int units = (B + 15) / 16;
    hipLaunchKernelGGL(train_tail_kernel, dim3(units), dim3(256), tail_lds_bytes(s), st, s, b, perm, B, inv_div);
}

// ------------------------------------------------------------------ train_grad1 ------
// grid = Fp/16 feature tiles; workgroup = nt1 waves (one per 16-wide output tile).
// Each wave: Gt[f][o] = sum_n xhat[n][f] * delta1[n][o] over the whole batch (f32 MFMA).
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

__global__ __launch_bounds__(640) void train_grad1_kernel(TrainShape s, TrainBuffers b, const int* __restrict__ perm, int B,
                                                          int nparts) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int H1 = s.widths[1], F = s.widths[0];
    const int H1p = (H1 + 15) & ~15;
    const int nt1 = H1p / 16;
    float* Dsh = sm;                       // [H1p]
    float* red = sm + H1p;                 // [nt1][16][2]
    const int tid = threadIdx.x, lane = tid & 63, t = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int ft = blockIdx.x;
    const long long PS = part_size(s);
    const bool ln = s.layernorm != 0;
    // phase 0: D (needed by every workgroup when LayerNorm is present)
    if (ln) {
        for (int o = tid; o < H1p; o += blockDim.x) {
            float acc = 0.f;
            if (o < H1)
                for (int w = 0; w < nparts; ++w) acc += b.part[(long long)w * PS + 1 + o];
            Dsh[o] = acc;
        }
    }
    // phase 0b: reduce the partial vectors of the small gradients (disjoint elements per thread)
    {
        const long long gt = (long long)blockIdx.x * blockDim.x + tid;
        const long long gn = (long long)gridDim.x * blockDim.x;
        for (long long e = gt; e < PS; e += gn) {
            float acc = 0.f;
            for (int w = 0; w < nparts; ++w) acc += b.part[(long long)w * PS + e];
            const long long gi = part_to_grad(s, e);
            if (gi >= 0) b.grad[gi] = acc;
        }
    }
    // main loop over the batch
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int Bpad = ((B + 15) / 16) * 16;
    const float* Xc = b.xhat_xs + ft * 16 + li;
    const float* Dc = b.delta1 + t * 16 + li;
    for (int n0 = 0; n0 < Bpad; n0 += 16) {
        float av[4], bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nr = n0 + 4 * j + lq;
            const int nc = nr < B ? nr : B - 1;
            const long long rr = perm ? perm[nc] : nc;
            av[j] = Xc[rr * (long long)b.Fp];
            bv[j] = Dc[(long long)nr * H1p];
        }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc1, 0, 0, 0);
    }
    __syncthreads();                       // Dsh visible
    const int o = t * 16 + li;
    const float* W1 = b.theta + s.off_W[0];
    const float* g = ln ? b.theta + s.off_lns : nullptr;
    const float* beta = ln ? b.theta + s.off_lnb : nullptr;
    float* gW1 = b.grad + s.off_W[0];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = ft * 16 + lq * 4 + r;
        const float G = acc0[r] + acc1[r];
        float pg = 0.f, pb = 0.f;
        if (o < H1 && f < F) {
            if (ln) {
                const float w = W1[o + (long long)f * H1];
                const float Do = Dsh[o];
                gW1[o + (long long)f * H1] = g[f] * G + beta[f] * Do;
                pg = w * G;
                pb = w * Do;
            } else {
                gW1[o + (long long)f * H1] = G;
            }
        }
        if (ln) {
#pragma unroll
            for (int sft = 8; sft > 0; sft >>= 1) {
                pg += __shfl_xor(pg, sft, WAVE);
                pb += __shfl_xor(pb, sft, WAVE);
            }
            if (li == 0) {
                red[(t * 16 + lq * 4 + r) * 2 + 0] = pg;
                red[(t * 16 + lq * 4 + r) * 2 + 1] = pb;
            }
        }
    }
    if (ln) {
        __syncthreads();
        if (tid < 16) {
            const int f = ft * 16 + tid;
            if (f < F) {
                float sg = 0.f, sb = 0.f;
                for (int w = 0; w < nt1; ++w) { sg += red[(w * 16 + tid) * 2]; sb += red[(w * 16 + tid) * 2 + 1]; }
                b.grad[s.off_lns + f] = sg;
                b.grad[s.off_lnb + f] = sb;
            }
        }
    }
}

void isk_launch_train_grad1(const TrainShape& s, const TrainBuffers& b, const int* perm, int B, int, hipStream_t st) {
    const int H1p = isk_pad16(s.widths[1]);
    const int nt1 = H1p / 16;
    const int nparts = (B + 15) / 16;
    const size_t lds = (size_t)(H1p + nt1 * 16 * 2) * sizeof(float);
    hipLaunchKernelGGL(train_grad1_kernel, dim3(b.Fp / 16), dim3(64 * nt1), lds, st, s, b, perm, B, nparts);
}

// ------------------------------------------------------------------ opt_step ---------
// Optimisers.jl 0.4.7: OptimiserChain(WeightDecay(reg), Adam | Nesterov), one thread per parameter.
// The batch loss sits in grad[n]; a non-finite loss (or an earlier one in this epoch) skips the
// update, as the reference throws before update! (src/iso.jl:186-189).
__global__ __launch_bounds__(256) void opt_step_kernel(OptArgs a, double* loss_acc) {
    const float lstep = a.grad[a.n];
    const bool bad = (loss_acc[1] != 0.0) || !isfinite(lstep);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (bad) loss_acc[1] = 1.0;
        else loss_acc[0] += (double)lstep;
    }
    if (bad) return;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256) {
        const float th = a.theta[i];
        const float g = a.grad[i] * a.gscale + a.reg * th;
        float d;
        if (a.kind == ISK_OPT_ADAM) {
            const float m = a.beta1 * a.m[i] + (1.f - a.beta1) * g;
            const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
            a.m[i] = m;
            a.v[i] = v;
            d = m / a.bc1 / (sqrtf(v / a.bc2) + a.eps) * a.lr;
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
