// Layer-by-layer ("layered") path for networks too wide for the fused kernels (first hidden width > 160
// or tail layers that do not fit LDS, e.g. densenet(layers=[F,512,512,512,1])).  Correctness-first:
// activations live in HBM between layers; the GEMMs reuse the f32 / split-bf16 forward kernels in
// dense-block mode and train_fwd1_kernel, plus the small kernels below.
//   forward   z_l = W_l a_{l-1} + b_l  (mlp_fwd kernels, dense mode) ... a_l = act(z_l)
//   loss      delta_L = 2 (out - y) w^2 / B .* act'(z_L)           (src/iso.jl:185,191)
//   backward  dW_l = delta_l' a_{l-1},  db_l = sum_n delta_l,  delta_{l-1} = (delta_l W_l) .* act'(z_{l-1})
#include "isk_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define WAVE 64

__device__ __forceinline__ float l_sigmoid(float z) {
    float e = __expf(-fabsf(z));
    float r = 1.0f / (1.0f + e);
    return z >= 0.f ? r : e * r;
}
__device__ __forceinline__ float l_act(int act, float z) {
    switch (act) {
        case ISK_ACT_SIGMOID: return l_sigmoid(z);
        case ISK_ACT_TANH: return tanhf(z);
        case ISK_ACT_RELU: return fmaxf(z, 0.f);
        case ISK_ACT_SWISH: return z * l_sigmoid(z);
        default: return z;
    }
}
__device__ __forceinline__ float l_dact(int act, float z) {
    switch (act) {
        case ISK_ACT_SIGMOID: { float a = l_sigmoid(z); return a * (1.f - a); }
        case ISK_ACT_TANH: { float a = tanhf(z); return 1.f - a * a; }
        case ISK_ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case ISK_ACT_SWISH: { float s = l_sigmoid(z); return s + z * s * (1.f - s); }
        default: return 1.f;
    }
}

__global__ __launch_bounds__(256) void act_kernel(const float* __restrict__ z, float* __restrict__ a, long long rows, int ld,
                                                  int width, int act) {
    const long long tot = rows * ld;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % ld);
        a[e] = (c < width) ? l_act(act, z[e]) : 0.f;
    }
}
void isk_launch_act(const float* z, float* a, long long rows, int ld, int width, int act, hipStream_t st) {
    long long b = (rows * ld + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(act_kernel, dim3((unsigned)b), dim3(256), 0, st, z, a, rows, ld, width, act);
}

__global__ __launch_bounds__(256) void dact_mul_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                                       float* __restrict__ d, long long rows, int ld, int width, int act) {
    const long long tot = rows * ld;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % ld);
        d[e] = (c < width) ? dz[e] * l_dact(act, z[e]) : 0.f;
    }
}
void isk_launch_dact_mul(const float* dz, const float* z, float* d, long long rows, int ld, int width, int act, hipStream_t st) {
    long long b = (rows * ld + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(dact_mul_kernel, dim3((unsigned)b), dim3(256), 0, st, dz, z, d, rows, ld, width, act);
}

// single workgroup: the batch loss is one number and B*nout is small next to the GEMMs of this path
__global__ __launch_bounds__(1024) void loss_delta_kernel(const float* __restrict__ z, int ld, const float* __restrict__ target,
                                                          const float* __restrict__ lossw, const int* __restrict__ perm, int B,
                                                          int Bpad, int nout, float inv_div, int lastact, float* __restrict__ delta,
                                                          int delta_ld, float* __restrict__ loss_slot, const int* __restrict__ rank_err) {
    __shared__ float red[16];
    float ls = 0.f;
    const long long tot = (long long)Bpad * delta_ld;
    // four elements per thread in flight: the row index -> target chain is two dependent loads
    for (long long e0 = threadIdx.x; e0 < tot; e0 += 4 * 1024) {
        float zz[4], tg[4];
        bool live[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long e = e0 + u * 1024;
            const int n = (int)(e / delta_ld), o = (int)(e % delta_ld);
            live[u] = e < tot && n < B && o < nout;
            zz[u] = 0.f; tg[u] = 0.f;
            if (live[u]) {
                const long long prow = perm ? perm[n] : n;
                zz[u] = z[(long long)n * ld + o];
                tg[u] = target[prow * nout + o];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long e = e0 + u * 1024;
            if (e >= tot) continue;
            float d = 0.f;
            if (live[u]) {
                const int o = (int)(e % delta_ld);
                const float out = l_act(lastact, zz[u]);
                const float w = lossw[o];
                const float r = (out - tg[u]) * w;
                ls += r * r;
                d = 2.f * r * w * inv_div * l_dact(lastact, zz[u]);
            }
            delta[e] = d;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ls += __shfl_xor(ls, o, WAVE);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ls;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += red[i];
        loss_slot[0] = t;
        loss_slot[1] = rank_err ? (float)*rank_err : 0.f;
    }
}
void isk_launch_loss_delta(const float* z, int ld, const float* target, const float* lossw, const int* perm, int B, int nout,
                           float inv_div, int lastact, float* delta, int delta_ld, float* loss_slot, const int* rank_err, hipStream_t st) {
    const int Bpad = ((B + 15) / 16) * 16;
    hipLaunchKernelGGL(loss_delta_kernel, dim3(1), dim3(1024), 0, st, z, ld, target, lossw, perm, B, Bpad, nout, inv_div, lastact,
                       delta, delta_ld, loss_slot, rank_err);
}

// cotangent seed of the coordinate pullback: delta_L[n][o] = cot[n][o] * lastact'(z_L[n][o]), zero padded
__global__ __launch_bounds__(256) void cot_delta_kernel(const float* __restrict__ z, int ld, const float* __restrict__ cot, long long B,
                                                        long long Bpad, int nout, int lastact, float* __restrict__ delta, int delta_ld) {
    const long long tot = Bpad * delta_ld;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long long)gridDim.x * 256) {
        const long long n = e / delta_ld;
        const int o = (int)(e % delta_ld);
        delta[e] = (n < B && o < nout) ? cot[n * nout + o] * l_dact(lastact, z[n * ld + o]) : 0.f;
    }
}
void isk_launch_cot_delta(const float* z, int ld, const float* cot, long long B, int nout, int lastact, float* delta, int delta_ld,
                          hipStream_t st) {
    const long long Bpad = ((B + 15) / 16) * 16;
    long long blocks = (Bpad * delta_ld + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cot_delta_kernel, dim3((unsigned)blocks), dim3(256), 0, st, z, ld, cot, B, Bpad, nout, lastact, delta, delta_ld);
}

// dW tile: block = (16-feature tile kt, 16-output tile ot), 4 waves split the batch rows, f32 MFMA,
// LDS combine.  D[k_local][o_local] = sum_n x[n][k] * delta[n][o];  stored gW[o + k*N].
// Blocks with kt == 0 also produce db[o] = sum_n delta[n][o] from the delta fragments they already hold.
__global__ __launch_bounds__(256) void dense_dw_kernel(const float* __restrict__ x, int x_ld, const int* __restrict__ perm,
                                                       const float* __restrict__ delta, int delta_ld, int B, int K, int N,
                                                       float* __restrict__ gW, float* __restrict__ gb) {
    __shared__ float red[4][256];
    __shared__ float dred[4][16];
    const int kt = blockIdx.x, ot = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int nchunks = (B + 15) / 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float dsum = 0.f;
    const float* xc = x + kt * 16 + li;
    const float* dc = delta + ot * 16 + li;
    for (int c0 = wave; c0 < nchunks; c0 += 16) {             // 4 chunks per trip: their loads are issued together
        float av[4][4], bv[4][4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int chunk = c0 + 4 * cb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = chunk * 16 + 4 * lq + j;
                const bool ok = chunk < nchunks && r < B;
                const long long rr = ok ? (perm ? perm[r] : r) : 0;
                av[cb][j] = ok ? xc[rr * (long long)x_ld] : 0.f;
                bv[cb][j] = ok ? dc[(long long)r * delta_ld] : 0.f;
            }
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cb][j], bv[cb][j], acc, 0, 0, 0);
                dsum += bv[cb][j];
            }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][lane * 4 + r] = acc[r];
    dsum += __shfl_xor(dsum, 16, WAVE);
    dsum += __shfl_xor(dsum, 32, WAVE);
    if (lq == 0) dred[wave][li] = dsum;
    __syncthreads();
    if (wave == 0) {
        const int o = ot * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = kt * 16 + lq * 4 + r;
            const float v = red[0][lane * 4 + r] + red[1][lane * 4 + r] + red[2][lane * 4 + r] + red[3][lane * 4 + r];
            if (o < N && k < K) gW[o + (long long)k * N] = v;
        }
        if (kt == 0 && lq == 0 && o < N && gb) gb[o] = dred[0][li] + dred[1][li] + dred[2][li] + dred[3][li];
    }
}
void isk_launch_dense_dw(const float* x, int x_ld, const int* perm, const float* delta, int delta_ld, int B, int K, int N,
                         float* gW, float* gb, hipStream_t st) {
    hipLaunchKernelGGL(dense_dw_kernel, dim3((K + 15) / 16, (N + 15) / 16), dim3(256), 0, st, x, x_ld, perm, delta, delta_ld, B, K, N,
                       gW, gb);
}

// WtT: fragment-linear image of W' so that  dz[n][k] = sum_o delta[n][o] W[o][k]  runs on train_fwd1_kernel:
// "input" dimension = o (padded to 16 chunks), "output" tiles over k.
__global__ __launch_bounds__(256) void fold_T_kernel(const float* __restrict__ theta, long long off_W, int K, int N,
                                                     float* __restrict__ WtT) {
    const int Np = (N + 15) & ~15;                 // contraction (chunks over o)
    const int Kp = (K + 15) & ~15;                 // output tiles over k
    const int nt = Kp / 16;
    const long long total = (long long)Np * Kp;
    const float* W = theta + off_W;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int j = e & 3;
        const int lane = (e >> 2) & 63;
        const long long ct = e >> 8;
        const int t = (int)(ct % nt);
        const int c = (int)(ct / nt);
        const int k = t * 16 + (lane & 15);
        const int o = c * 16 + 4 * (lane >> 4) + j;
        WtT[e] = (o < N && k < K) ? W[o + (long long)k * N] : 0.f;
    }
}
void isk_launch_fold_T(const float* theta, long long off_W, int K, int N, float* WtT, hipStream_t st) {
    long long total = (long long)isk_pad16(K) * isk_pad16(N);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fold_T_kernel, dim3(blocks), dim3(256), 0, st, theta, off_W, K, N, WtT);
}

// one wave per feature row f: dW1[o][f] = g[f] G[o][f] + beta[f] D[o];  dg[f] = sum_o W1 G;  dbeta[f] = sum_o W1 D
__global__ __launch_bounds__(256) void ln_unfold_kernel(const float* __restrict__ G, const float* __restrict__ D,
                                                        const float* __restrict__ theta, long long off_W, long long off_lns,
                                                        long long off_lnb, int F, int H1, float* __restrict__ grad) {
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (f >= F) return;
    const float gf = theta[off_lns + f], bf = theta[off_lnb + f];
    float pg = 0.f, pb = 0.f;
    for (int o = lane; o < H1; o += WAVE) {
        const float g0 = G[o + (long long)f * H1];
        const float w = theta[off_W + o + (long long)f * H1];
        const float d = D[o];
        grad[off_W + o + (long long)f * H1] = gf * g0 + bf * d;
        pg += w * g0;
        pb += w * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { pg += __shfl_xor(pg, o, WAVE); pb += __shfl_xor(pb, o, WAVE); }
    if (lane == 0) { grad[off_lns + f] = pg; grad[off_lnb + f] = pb; }
}
void isk_launch_ln_unfold(const float* G, const float* D, const float* theta, long long off_W, long long off_lns, long long off_lnb,
                          int F, int H1, float* grad, hipStream_t st) {
    hipLaunchKernelGGL(ln_unfold_kernel, dim3((F + 3) / 4), dim3(256), 0, st, G, D, theta, off_W, off_lns, off_lnb, F, H1, grad);
}
