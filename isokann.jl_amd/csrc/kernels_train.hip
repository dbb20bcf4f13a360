// Training-path kernels for gfx950 (MI355X): one optimizer step of train_batch!
// (src/iso.jl:179-194 of the reference) is the chain
//   train_fwd1n -> K-slice partials of z1 = W1 * (g .* xhat[batch] + beta)   (f32 MFMA; workgroup = 32 rows x one K slice)
//   train_tail  -> z1 = sum of the partials + b1, tail layers fwd, weighted loss, backward to delta1
//                  (VALU, 16 samples / workgroup), per-workgroup partial sums of the small gradients
//   train_grad1n-> G = delta1' * xhat[batch]  (f32 MFMA), LayerNorm-affine unfold:
//                  dW1 = g.*G + beta*D', dg = sum_o W1.*G, dbeta = W1'*D; reduces the partials
//   opt_step    -> WeightDecay + Adam | Nesterov on the flat parameter buffer (Optimisers.jl rules)
// The LayerNorm standardisation has no trainable parameters, so it is hoisted out of the loop
// (xhat is cached); its affine part (g, beta) is applied to the sample operand as it is loaded and unfolded
// on the way back, which removes the dL/d(a0) GEMM entirely: only G (one GEMM) is needed.
//
// "Native" parameter layout (fused path): inside the engine the layer-1 weights live in the FRAGMENT-LINEAR order the
// MFMA B operand wants -- W1n[((c*nt + t)*64 + lane)*4 + j] = W1[o = 16t + lane%16][k = 16c + 4(lane/16) + j], zero padded to
// [Fp/16][nt] whole fragments -- and so do their gradient and optimizer state (the optimizer is elementwise, any order
// does).  The LayerNorm scale / bias blocks are padded to Fp.  Consequences: no per-step re-layout ("fold") kernel at all,
// train_fwd1n's weight loads and train_grad1n's gradient stores are whole coalesced KiB straight from / to the
// accumulator registers.  The Flux order exists only at the C ABI (isk_get/set_params convert on the host).
#include <cstdlib>
#include "isk_internal.h"
#include "train_fwd1n.hpp"
#include "opt_rule.hpp"

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
// NOT inlined: train_tail calls these from a dozen places and each inlined copy carries every activation's code (tanhf
// alone is ~100 instructions); the kernel is a chain of short phases that each START with instruction-cache misses, so
// its size is its latency (8280 -> see DESIGN.md instructions)
__device__ __attribute__((noinline)) float t_act(int act, float z) {
    switch (act) {
        case ISK_ACT_SIGMOID: return t_sigmoid(z);
        case ISK_ACT_TANH: return tanhf(z);
        case ISK_ACT_RELU: return fmaxf(z, 0.f);
        case ISK_ACT_SWISH: return z * t_sigmoid(z);
        default: return z;
    }
}
__device__ __attribute__((noinline)) float t_dact(int act, float z, float a) {
    switch (act) {
        case ISK_ACT_SIGMOID: return a * (1.f - a);
        case ISK_ACT_TANH: return 1.f - a * a;
        case ISK_ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case ISK_ACT_SWISH: { float s = t_sigmoid(z); return s + a * (1.f - s); }
        default: return 1.f;
    }
}

// diagnostic build -DISK_KSTAMP: every training kernel's first workgroup appends (kernel id, s_memrealtime) to a ring --
// the start times of consecutive launches WITHOUT a profiler attached (scripts/kstamps.py)
#ifdef ISK_KSTAMP
__device__ unsigned long long g_kstamps[2048];
__device__ unsigned g_kstamp_n;
extern "C" void isk_debug_kstamps(unsigned long long* out, unsigned* n) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kstamps), sizeof(unsigned long long) * 2048);
    (void)hipMemcpyFromSymbol(n, HIP_SYMBOL(g_kstamp_n), sizeof(unsigned));
}
#define KSTAMP(id) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { \
        const unsigned k_ = atomicAdd(&g_kstamp_n, 1u); \
        g_kstamps[k_ & 2047] = ((unsigned long long)(id) << 56) | (__builtin_amdgcn_s_memrealtime() & 0xffffffffffffffull); } } while (0)
#else
#define KSTAMP(id) do { } while (0)
#endif

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

// generic row GEMM on the same kernel: out[n][16*t + j] = sum_k X[row(n)][k] * Wt(k, t) + bias, any layer
void isk_launch_gemm_rows(const float* X, int ldx, const float* Wt, const float* bias, int nt, const int* perm, int B, float* out,
                          hipStream_t st) {
    const int units = (B + 15) / 16;
    const int upx = (units + 7) / 8;
    hipLaunchKernelGGL(train_fwd1_kernel, dim3(8 * upx * nt), dim3(FW1_WAVES * 64), 0, st, X, ldx, ldx / 16, Wt, bias, nt, perm, B, out);
}


// ------------------------------------------------------------------ train_fwd1n ------
// grid (row blocks of 32 batch rows, tile groups x KS K-slices); workgroup = 8 waves that split the slice's 16-feature
// chunks; every wave accumulates the group's NTB (<= 5) output tiles of both 16-row units, the eight partial tile sets
// are combined through LDS and written as ONE K-slice partial zp[ks][row][H1p] (train_tail sums the KS partials as it
// loads them).  Sized to run BESIDE the side-stream featurizer, whose waves hold ~248 of a SIMD's 512 registers for the
// whole training phase: <= 128 registers per wave (two workgroups per CU fit next to it), at most two chunks per wave so
// that all of a wave's loads are in flight before its first MFMA (one memory round trip per launch).  At C3 (B = 1000):
// 32 row blocks x 2 tile groups x 8 slices = 512 workgroups of 100 KB input each = 51 MB through L2 per step, against
// 125 MB for the 567 (unit, tile) workgroups of the first version.
template <int NTB, bool LN>
__global__ __launch_bounds__(FW1N_WAVES * 64, 4) void train_fwd1n_kernel(Fwd1nArgs a) {
    ISK_LATENCY_CHAIN_PRIO();
    KSTAMP(1);
    extern __shared__ __attribute__((aligned(16))) float red[];          // [4 wave pairs][2 units][NTB][256]: 40 KB at NTB = 5
    fwd1n_body<NTB, LN, false>(a, red);
}

// K-slice plan of train_fwd1n for a batch of B rows and nt layer-1 output tiles: about two workgroups per CU, at least
// one chunk per wave
void isk_fwd1n_plan(int B, int Fp, int nt, int* KS, int* cpb) {
    const int nch = Fp / 16, RB = (B + 31) / 32, ntg = (nt + FW1N_NTB - 1) / FW1N_NTB;
    int ks = 512 / (RB * ntg);
    if (ks > ISK_FWD1_MAXKS) ks = ISK_FWD1_MAXKS;
    if (ks > nch / FW1N_WAVES) ks = nch / FW1N_WAVES;
    if (ks < 1) ks = 1;
    *cpb = (nch + ks - 1) / ks;
    *KS = (nch + *cpb - 1) / *cpb;
}

void isk_launch_train_fwd1n(const TrainShape& s, const TrainBuffers& b, const int* perm, int B, hipStream_t st) {
    Fwd1nArgs a;
    a.X = b.xhat_xs; a.Fp = b.Fp; a.nch = b.Fp / 16; a.perm = perm; a.B = B;
    a.W1n = b.theta + s.off_W[0];
    a.g = s.layernorm ? b.theta + s.off_lns : nullptr;
    a.beta = s.layernorm ? b.theta + s.off_lnb : nullptr;
    a.nt = isk_pad16(s.widths[1]) / 16;
    a.ntg = (a.nt + FW1N_NTB - 1) / FW1N_NTB;
    isk_fwd1n_plan(B, b.Fp, a.nt, &a.KS, &a.cpb);
    a.zp = b.a1; a.zstride = b.zstride;
    const int RB = (B + 31) / 32;
    // tiles per group: the groups are as even as the tile count allows (9 tiles: 5 + 4)
    const int ntb = (a.nt + a.ntg - 1) / a.ntg;
    const dim3 grid(RB, a.ntg * a.KS), blk(FW1N_WAVES * 64);
#define FCASE(N_)                                                                                                           \
    case N_: {                                                                                                              \
        const size_t lds = (size_t)4 * 2 * N_ * 256 * sizeof(float);                                                        \
        if (a.g) {                                                                                                          \
            if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)train_fwd1n_kernel<N_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
            hipLaunchKernelGGL((train_fwd1n_kernel<N_, true>), grid, blk, lds, st, a);                                      \
        } else {                                                                                                            \
            if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)train_fwd1n_kernel<N_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((train_fwd1n_kernel<N_, false>), grid, blk, lds, st, a);                                     \
        }                                                                                                                   \
    } break;
    switch (ntb) { FCASE(1) FCASE(2) FCASE(3) FCASE(4) default: FCASE(5) }
#undef FCASE
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
// ... | b_1 | b_2 | ... behind the weights: the tail biases (a global load in the middle of a phase is a full memory round trip)
__host__ __device__ static inline int tail_off_b(const TrainShape& s, int l) {
    int o = tail_off_w(s, s.nlayers);
    for (int i = 1; i < l; ++i) o += s.widths[i + 1];
    return o;
}

// one 16 x 16 tile  C[m][n] = sum_k A(m, k) * B(k, n)  on the f32 MFMA, operands fetched element-wise (LDS): fa(k) is
// this lane's A(lane % 16, k), fb(k) its B(k, lane % 16); step s contracts k = 4 s + lane / 16.  Four independent
// accumulation chains hide the MFMA latency; callers mask k >= K (and columns past the matrix) to 0 on BOTH operands.
template <class FA, class FB>
__device__ __forceinline__ f32x4 tail_tile_mma(int st0, int nsteps, int lq, FA fa, FB fb) {
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
    int st = st0;
    for (; st + 3 < nsteps; st += 4) {
        const int k0 = 4 * st + lq;
        const float a0 = fa(k0), a1 = fa(k0 + 4), a2 = fa(k0 + 8), a3 = fa(k0 + 12);
        const float b0 = fb(k0), b1 = fb(k0 + 4), b2 = fb(k0 + 8), b3 = fb(k0 + 12);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, c3, 0, 0, 0);
    }
    for (; st < nsteps; ++st) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa(4 * st + lq), fb(4 * st + lq), c0, 0, 0, 0);
    return (c0 + c1) + (c2 + c3);
}

#ifdef ISK_TAIL_STAMP
__device__ unsigned long long g_tail_stamps[32];
extern "C" void isk_debug_tail_stamps(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_stamps), sizeof(unsigned long long) * 32); }
#define TSTAMP(k) do { if (blockIdx.x == 7 && threadIdx.x == 0) { g_tail_stamps[k] = __builtin_amdgcn_s_memrealtime(); g_tail_stamps[16 + k] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define TSTAMP(k) do { } while (0)
#endif

// The kernel is a chain of short dependent phases on 16 samples: what it costs is latency, and in its first versions most
// of that was SCALAR: TrainShape arrived by value and every phase re-derived its LDS offsets, tile counts and loop bounds
// from it with chains of dependent kernarg loads and integer divisions (in-kernel stamps: 1 us of "setup" in front of
// each 0.9 us of arithmetic).  So: the host precomputes one descriptor per tail layer (TailLay, fetched as a block),
// element loops are 2-D (row = tid / 32, column = tid % 32 + 32 i: no division), the products run as MFMA tiles.
// 512 threads at <= 128 registers: two waves per SIMD, which is what fits beside a featurizer wave (~248 of the SIMD's
// 512 registers).
#define TAIL_THREADS 512
#define TAIL_ZI 5                          /* z1 columns per thread: H1 <= 160 on the fused path (ISK_MAX_NT tiles) */
#define TAIL_WU (2048 / TAIL_THREADS)      /* tail weights per thread and layer in the prefetched first pass */
struct TailLay {
    int K, N, act;              // inputs, outputs, activation of this layer
    int tN, tK;                 // 16-wide tiles over N and over K
    int nsteps, ksplit, spp;    // forward contraction: 4-deep steps, wave split, steps per part
    int oAp, oZp;               // LDS float offsets: previous layer's activations / pre-activations (delta after the backward)
    int oA, oZ, oW, oB;         //                    this layer's activations, pre-activations, weights (Flux), bias
    int nW;                     // K * N
    long long gW, gb;           // offsets in theta
    long long pW, pb;           // offsets in the per-workgroup partial vector
};
struct TailDesc {
    int L;                      // tail layers (model layers - 1)
    int H1, H1p, nout, act0, lastact, hidden_act;
    int oP;                     // LDS offset of the [8][256] partial-tile buffer
    long long off_b0, PS;       // layer-1 bias in theta; partial-vector length
    TailLay lay[ISK_MAX_LAYERS - 1];
};

__global__ __launch_bounds__(TAIL_THREADS, 4) void train_tail_kernel(TailDesc d, TrainBuffers b, const int* __restrict__ perm,
                                                                  int B, float inv_div) {
    ISK_LATENCY_CHAIN_PRIO();
    KSTAMP(2);
    TSTAMP(0);
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float lred[TAIL_THREADS / 64];
    constexpr int NTH = TAIL_THREADS, NWV = NTH / 64;
    const int L = d.L;
    const int tid = threadIdx.x;
    const int unit = blockIdx.x;
    const int H1 = d.H1, H1p = d.H1p, nout = d.nout;
    const int wv = tid >> 6, li = tid & 15, lq = (tid >> 4) & 3;
    const int r16 = tid >> 5, c32 = tid & 31;                   // 2-D element mapping: sample row, column (+ 32 i)
    // ---- prologue: everything the first pass of every phase needs from memory is requested here, before the first wait
    const int row_g = unit * 16 + r16;
    const bool rvalid = row_g < B;
    const long long lp_row = perm ? perm[rvalid ? row_g : B - 1] : (rvalid ? row_g : B - 1);     // this thread's sample
    float zv[TAIL_ZI];
    {
        const float* b1 = b.theta + d.off_b0;
        const float* zrow = b.a1 + (long long)row_g * H1p;
#pragma unroll
        for (int i = 0; i < TAIL_ZI; ++i) {
            // z1 = b1 + the KS K-slice partials of train_fwd1n: branch-free and fully unrolled (clamped addresses, masked
            // adds) -- as a run-time loop every partial was a load -> wait -> add round trip of its own
            const int c = c32 + 32 * i;
            const int cc = c < H1 ? c : H1 - 1;
            float pz[ISK_FWD1_MAXKS];
#pragma unroll
            for (int ks = 0; ks < ISK_FWD1_MAXKS; ++ks) pz[ks] = zrow[(long long)(ks < b.KS ? ks : 0) * b.zstride + cc];
            float z = b1[cc];
#pragma unroll
            for (int ks = 0; ks < ISK_FWD1_MAXKS; ++ks) z += ks < b.KS ? pz[ks] : 0.f;
            zv[i] = z;
        }
    }
    constexpr int TAIL_PF = 2;          // tail layers whose weights / biases are prefetched (deeper tails: plain loops below)
    float wpre[TAIL_PF][TAIL_WU], bpre[TAIL_PF];
#pragma unroll
    for (int i = 0; i < TAIL_PF; ++i) {
        const TailLay y = d.lay[i < L ? i : (L > 0 ? L - 1 : 0)];    // a duplicate of the last layer when the tail is shorter
        const float* Wg = b.theta + y.gW;
#pragma unroll
        for (int u = 0; u < TAIL_WU; ++u) { const int e = u * NTH + tid; wpre[i][u] = Wg[e < y.nW ? e : (y.nW > 0 ? y.nW - 1 : 0)]; }
        bpre[i] = b.theta[y.gb + (tid < y.N ? tid : 0)];
    }
    const bool lp = c32 < nout;                          // loss phase: thread (r16, c32) owns output c32 of sample r16 (nout <= 32)
    const float lp_tgt = lp ? b.target[lp_row * nout + c32] : 0.f;
    const float lp_w = (lp && !b.cot_mode) ? b.lossw[c32] : 0.f;
    // Adam's bias corrections, kept on the device so that a step's launch arguments never change (hipGraph replay):
    // bt[0..1] = beta^t of the coming optimizer step; handed to opt_step in bt[2..3] and advanced at the END of the kernel
    const bool bt_owner = b.bt && unit == 0 && tid == 0;
    double bt0 = 0.0, bt1 = 0.0;
    if (bt_owner) { bt0 = b.bt[0]; bt1 = b.bt[1]; }
    // ---- LDS staging: Z_0 / A_0 (16 x H1), tail weights (Flux layout W[o + k*N]) and biases
    {
        float* A0 = sm;
        float* Z0 = sm + 16 * H1;
#pragma unroll
        for (int i = 0; i < TAIL_ZI; ++i) {
            const int c = c32 + 32 * i;
            if (c < H1) { Z0[r16 * H1 + c] = zv[i]; A0[r16 * H1 + c] = t_act(d.act0, zv[i]); }
        }
    }
#pragma unroll
    for (int i = 0; i < TAIL_PF; ++i) {
        if (i < L) {
            const TailLay y = d.lay[i];
            float* Wl = sm + y.oW;
#pragma unroll
            for (int u = 0; u < TAIL_WU; ++u) { const int e = u * NTH + tid; if (e < y.nW) Wl[e] = wpre[i][u]; }
            for (int e = TAIL_WU * NTH + tid; e < y.nW; e += NTH) Wl[e] = b.theta[y.gW + e];      // > 2048 weights in the layer
            if (tid < y.N) sm[y.oB + tid] = bpre[i];
            for (int e = NTH + tid; e < y.N; e += NTH) sm[y.oB + e] = b.theta[y.gb + e];
        }
    }
    for (int l = TAIL_PF; l < L; ++l) {                                   // deeper tails: not prefetched
        const TailLay y = d.lay[l];
        for (int e = tid; e < y.nW; e += NTH) sm[y.oW + e] = b.theta[y.gW + e];
        for (int e = tid; e < y.N; e += NTH) sm[y.oB + e] = b.theta[y.gb + e];
    }
    __syncthreads();
    TSTAMP(1);
    float* pbuf = sm + d.oP;                                              // [NWV][256] partial tiles
    // ---- forward through the tail: Z_l (16 x N) = A_{l-1} (16 x K) * W_l' on the f32 MFMA.  A wave per 16-column tile;
    // narrow layers (fewer tiles than waves: the C3 tail is ONE tile, 142 deep) also split the contraction over the waves
    for (int l = 0; l < L; ++l) {
        const TailLay y = d.lay[l];
        const int K = y.K, N = y.N;
        const float* arow = sm + y.oAp + li * K;
        const float* Wl = sm + y.oW;
        float* Al = sm + y.oA;
        float* Zl = sm + y.oZ;
        const float* bias = sm + y.oB;
        int t = 0, part = wv;                                             // job j = t * ksplit + part, dealt round-robin
        while (part >= y.ksplit) { part -= y.ksplit; ++t; }
        for (; t < y.tN;) {
            const int n = 16 * t + li;
            const bool nok = n < N;
            const float* wcol = Wl + (nok ? n : N - 1);
            int s1 = (part + 1) * y.spp;
            if (s1 > y.nsteps) s1 = y.nsteps;
            const f32x4 acc = tail_tile_mma(part * y.spp, s1, lq,
                [&](int k) { const float v = arow[k < K ? k : K - 1]; return k < K ? v : 0.f; },
                [&](int k) { const float v = wcol[(k < K ? k : K - 1) * N]; return (k < K && nok) ? v : 0.f; });
            if (y.ksplit > 1) {
                *(f32x4*)(pbuf + (wv * 64 + (tid & 63)) * 4) = acc;      // tN * ksplit <= NWV: job j runs on wave j
            } else if (nok) {
                const float bn = bias[n];
#pragma unroll
                for (int r = 0; r < 4; ++r) {                            // accumulator rows 4 lq + r = samples, column n
                    const float z = acc[r] + bn;
                    Zl[(4 * lq + r) * N + n] = z;
                    Al[(4 * lq + r) * N + n] = t_act(y.act, z);
                }
            }
            part += NWV;
            while (part >= y.ksplit) { part -= y.ksplit; ++t; }
        }
        if (y.ksplit > 1) {
            __syncthreads();
            for (int e = tid; e < y.tN * 256; e += NTH) {                 // element (tile, lane', r) of the accumulator layout
                const int tt = e >> 8, lp_ = (e >> 2) & 63, r = e & 3;
                const int n = 16 * tt + (lp_ & 15), row = 4 * (lp_ >> 4) + r;
                if (n < N) {
                    float z = bias[n];
                    for (int pp = 0; pp < y.ksplit; ++pp) z += pbuf[(tt * y.ksplit + pp) * 256 + (e & 255)];
                    Zl[row * N + n] = z;
                    Al[row * N + n] = t_act(y.act, z);
                }
            }
        }
        __syncthreads();
    }
    TSTAMP(2);
    // ---- loss + delta of the last layer (overwrites Z_{L-1}; with no tail layers that is Z_0)
    float lsum = 0.f;
    {
        float* AL = L > 0 ? sm + d.lay[L - 1].oA : sm;
        float* ZL = L > 0 ? sm + d.lay[L - 1].oZ : sm + 16 * H1;
        if (nout <= 32) {
            if (lp) {
                const int p = r16 * nout + c32;
                const float out = AL[p];
                float dl = 0.f;
                if (b.cot_mode) {                              // pullback: delta_L = cotangent .* act'(z_L)
                    if (rvalid) dl = lp_tgt * t_dact(d.lastact, ZL[p], out);
                } else {
                    const float r = (out - lp_tgt) * lp_w;
                    if (rvalid) {
                        lsum += r * r;
                        dl = 2.f * r * lp_w * inv_div * t_dact(d.lastact, ZL[p], out);
                    }
                }
                ZL[p] = dl;
            }
        } else {
            for (int p = tid; p < 16 * nout; p += NTH) {       // wide outputs: the plain loop
                const int sidx = p / nout, o = p - sidx * nout;
                const int row = unit * 16 + sidx;
                const bool valid = row < B;
                const int rowc = valid ? row : B - 1;
                const long long prow = perm ? perm[rowc] : rowc;
                const float tg = b.target[prow * nout + o];
                const float out = AL[p];
                float dl = 0.f;
                if (b.cot_mode) {
                    if (valid) dl = tg * t_dact(d.lastact, ZL[p], out);
                } else {
                    const float w = b.lossw[o];
                    const float r = (out - tg) * w;
                    if (valid) {
                        lsum += r * r;
                        dl = 2.f * r * w * inv_div * t_dact(d.lastact, ZL[p], out);
                    }
                }
                ZL[p] = dl;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o, WAVE);
    if ((tid & 63) == 0) lred[tid >> 6] = lsum;
    __syncthreads();
    TSTAMP(3);
    float* part_out = b.part + (long long)unit * d.PS;
    if (tid == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) t += lred[w];
        part_out[0] = t;
    }
    // ---- backward through the tail.  Both products of a layer as MFMA tile jobs dealt to the waves (they read A_{l-1},
    // delta_l, W_l and write the partial dW_l / Z_{l-1}: no barrier between them):
    //   dW_l (N x K)       = delta_l' (N x 16) * A_{l-1} (16 x K)          jobs [0, tN*tK)
    //   delta_{l-1} (16xK) = (delta_l (16 x N) * W_l (N x K)) .* act'      jobs [tN*tK, tN*tK + tK)
    for (int l = L - 1; l >= 0; --l) {
        const TailLay y = d.lay[l];
        const int K = y.K, N = y.N;
        const float* Ap = sm + y.oAp;
        float* Zp = sm + y.oZp;
        const float* Zl = sm + y.oZ;
        const float* Wl = sm + y.oW;
        float* pW = part_out + y.pW;
        float* pb = part_out + y.pb;
        const int pact = l > 0 ? d.hidden_act : d.act0;                  // activation that produced A_{l-1}
        const int njobs = y.tN * y.tK + y.tK;
        int to = 0, tk = wv;                                             // job (to, tk); to == tN: the delta_{l-1} jobs
        while (tk >= y.tK) { tk -= y.tK; ++to; }
        for (int j = wv; j < njobs; j += NWV) {
            const int kk = 16 * tk + li;                                 // input index of this lane's column
            const bool kok = kk < K;
            const int kc = kok ? kk : K - 1;
            if (to < y.tN) {
                const int o = 16 * to + li;
                const bool ook = o < N;
                const int oc = ook ? o : N - 1;
                const f32x4 acc = tail_tile_mma(0, 4, lq,
                    [&](int q) { const float v = Zl[q * N + oc]; return ook ? v : 0.f; },
                    [&](int q) { const float v = Ap[q * K + kc]; return kok ? v : 0.f; });
                if (kok) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                        // accumulator rows = outputs 16 to + 4 lq + r, column = input kk
                        const int oo = 16 * to + 4 * lq + r;
                        if (oo < N) pW[oo + (long long)kk * N] = acc[r];
                    }
                }
            } else {
                const f32x4 acc = tail_tile_mma(0, (N + 3) >> 2, lq,
                    [&](int o) { const float v = Zl[li * N + (o < N ? o : N - 1)]; return o < N ? v : 0.f; },
                    [&](int o) { const float v = Wl[(o < N ? o : N - 1) + kc * N]; return (o < N && kok) ? v : 0.f; });
                if (kok) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                        // rows = samples 4 lq + r, column = input kk; in place
                        const int p = (4 * lq + r) * K + kk;
                        Zp[p] = acc[r] * t_dact(pact, Zp[p], Ap[p]);
                    }
                }
            }
            tk += NWV;
            while (tk >= y.tK) { tk -= y.tK; ++to; }
        }
        for (int o = tid; o < N; o += NTH) {
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc += Zl[q * N + o];
            pb[o] = acc;
        }
        __syncthreads();
    }
    TSTAMP(4);
    // ---- delta1 -> global (zero padded), columns in the order train_grad1n loads them: one dwordx4 per row covers FOUR
    // output tiles (position 64 q + 4 c + u holds output 16 (4 q + u) + c), see isk_delta1_pos
    {
        const float* Z0 = sm + 16 * H1;
        float* drow = b.delta1 + (long long)row_g * H1p;
        for (int c = c32; c < H1p; c += 32) drow[isk_delta1_pos(c, H1p)] = (c < H1) ? Z0[r16 * H1 + c] : 0.f;
    }
    if (bt_owner) {
        b.bt[2] = bt0; b.bt[3] = bt1;
        b.bt[0] = bt0 * b.beta1; b.bt[1] = bt1 * b.beta2;
    }
    TSTAMP(5);
}

static TailDesc tail_desc(const TrainShape& s, size_t* lds_bytes) {
    TailDesc d = {};
    const int L = s.nlayers - 1;
    d.L = L; d.H1 = s.widths[1]; d.H1p = isk_pad16(s.widths[1]); d.nout = s.widths[s.nlayers];
    d.act0 = (s.nlayers == 1) ? s.lastact : s.act; d.lastact = s.lastact; d.hidden_act = s.act;
    d.off_b0 = s.off_b[0]; d.PS = part_size(s);
    long long poff = 1 + s.widths[1];
    for (int i = 0; i < L; ++i) {
        const int l = i + 1;                       // model layer index
        TailLay& y = d.lay[i];
        y.K = s.widths[l]; y.N = s.widths[l + 1];
        y.act = (l == s.nlayers - 1) ? s.lastact : s.act;
        y.tN = (y.N + 15) / 16; y.tK = (y.K + 15) / 16;
        y.nsteps = (y.K + 3) / 4;
        y.ksplit = y.tN < TAIL_THREADS / 64 ? (TAIL_THREADS / 64) / y.tN : 1;
        if (y.ksplit > y.nsteps) y.ksplit = y.nsteps;
        y.spp = (y.nsteps + y.ksplit - 1) / y.ksplit;
        y.oAp = tail_off_act(s, l - 1); y.oZp = y.oAp + 16 * y.K;
        y.oA = tail_off_act(s, l); y.oZ = y.oA + 16 * y.N;
        y.oW = tail_off_w(s, l); y.oB = tail_off_b(s, l);
        y.nW = y.K * y.N;
        y.gW = s.off_W[l]; y.gb = s.off_b[l];
        y.pW = poff; y.pb = poff + (long long)y.N * y.K;
        poff = y.pb + y.N;
    }
    d.oP = tail_off_b(s, s.nlayers);
    *lds_bytes = ((size_t)d.oP + (TAIL_THREADS / 64) * 256) * sizeof(float);
    return d;
}

void isk_launch_train_tail(const TrainShape& s, const TrainBuffers& b, const int* perm, int B, float inv_div,
                           hipStream_t st) {
    if (!b.tail_generic && isk_train_tail_narrow_ok(s)) { isk_launch_train_tail_narrow(s, b, perm, B, inv_div, st); return; }
    const int units = (B + 15) / 16;
    size_t lds;
    const TailDesc d = tail_desc(s, &lds);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)train_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(train_tail_kernel, dim3(units), dim3(TAIL_THREADS), lds, st, d, b, perm, B, inv_div);
}

// ------------------------------------------------------------------ train_grad1n -----
// grid = (Fp/16 feature tiles, row slabs); workgroup = NW waves that SPLIT THE BATCH ROWS (the step is
// latency-bound, so every SIMD gets several waves of independent gathers); each wave accumulates all
// NT output tiles:  Gt[f][o] = sum_n xhat[n][f] * delta1[n][o]  (f32 MFMA).  The NW partial tile sets are combined
// through LDS by ALL threads, and because the engine keeps W1 (and its gradient) fragment-linear, accumulator
// element (tile t, lane, r) IS gradient element ft*NT*256 + t*256 + 4*lane + r: the workgroup's share of dW1 is one
// contiguous range, read (W1, for the LayerNorm unfold) and written as whole coalesced lines, no transposition.
// delta1 rows are stored by train_tail in isk_delta1_pos order, so the dwordx4 a lane loads per row is column li of
// four consecutive tiles.
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


#ifdef ISK_TAIL_STAMP
#define GSTAMP(k) do { if (blockIdx.x == 5 && blockIdx.y == 0 && threadIdx.x == 0) g_tail_stamps[8 + k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GSTAMP(k) do { } while (0)
#endif
// OPT (one row slab, one GPU): the optimizer step rides on this kernel -- the workgroup that produces a gradient element owns the
// parameter.  Round 2 measured that fusion at -2 % (3.87 vs 3.96 ms per iteration at minibatch 100): its m / v loads sat at the END
// of the kernel, a dependent round trip that cost what the saved launch was worth.  Here every optimizer operand -- theta, m, v of
// the W1 share, of the LayerNorm / bias / tail-parameter elements this thread will update, the units' loss partials, the epoch's
// collapse flag, Adam's bias corrections -- is requested at the kernel's start and arrives under the main loop.
template <int NT, int NW, bool OPT = false>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 4 : 0) void train_grad1n_kernel(TrainShape s, TrainBuffers b, const int* __restrict__ perm,
                                                              int B, int nparts, OptArgs oa) {
    ISK_LATENCY_CHAIN_PRIO();
    KSTAMP(3);
    GSTAMP(0);
    constexpr int NB4 = NT / 4, NR = NT % 4;
    constexpr int CB = 4;                          // chunks per batch: their row indices are fetched together
    constexpr int NTH = NW * 64;
    static_assert(NTH % 256 == 0, "a thread keeps one (lane, r) of the accumulator layout across tiles");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* red = sm;                               // [NW][NT][256]
    float* Dred = sm + NW * NT * 256;              // [NW][NT][16]
    float* Dsh = Dred + NW * NT * 16;              // [NT*16]
    float* psh = Dsh + NT * 16;                    // [NW][2][4]  per-wave (dg, dbeta) partials of 4 features
    const int H1 = s.widths[1];
    const int H1p = NT * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int ft = blockIdx.x;
    const int slab = blockIdx.y, nslab = gridDim.y;     // batch rows are split over `nslab` workgroups per feature tile;
    float* const gout = b.grad + (long long)slab * (s.nparams + 2);   // each writes its (linear) share to its own gradient slab
    const long long PS = part_size(s);
    const bool ln = s.layernorm != 0;
    constexpr int NE = (NT * 256 + NTH - 1) / NTH;
    const int rr_ = tid & 3, lanep = (tid >> 2) & 63, lqp = lanep >> 4, lip = lanep & 15;
    const int k = ft * 16 + 4 * lqp + rr_;
    const float* W1n = b.theta + s.off_W[0] + (long long)ft * NT * 256;
    // small-gradient partial vectors: one element per 16-lane group, lanes split the partials (slab 0 only).  The loads of a
    // group's (first) element go out HERE, ahead of the main loop's own loads, and are summed behind it: done in front of the
    // loop the reduction was a memory round trip of its own for the first PS / 32 workgroups (2.2 us of the kernel's 15).
    const long long red_groups = (long long)gridDim.x * (NW * 4);
    const long long red_grp = (long long)blockIdx.x * (NW * 4) + (tid >> 4);
    const bool red_pre = slab == 0 && red_grp < PS && !(red_grp >= 1 && red_grp < 1 + H1);
    float red_pv[4] = {0.f, 0.f, 0.f, 0.f};
    if (red_pre) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int w = li + 16 * u; red_pv[u] = (w < nparts) ? b.part[(long long)w * PS + red_grp] : 0.f; }
    }
    // ---- OPT: every operand of the optimizer step this thread will take, requested now
    float o_th[NE], o_m[NE], o_v[NE];
    float b1_grad = 0.f;
    float ln_th = 0.f, ln_m = 0.f, ln_v = 0.f, b1_th = 0.f, b1_m = 0.f, b1_v = 0.f, sp_th = 0.f, sp_m = 0.f, sp_v = 0.f, lossp = 0.f;
    long long sp_gi = -1;
    float sp_racc = 0.f;
    double o_coll = 0.0;
    float bc1 = 1.f, bc2 = 1.f, o_peer = 0.f;
    const bool adam = oa.kind == ISK_OPT_ADAM;
    const long long wbase = s.off_W[0] + (long long)ft * NT * 256;
    GSTAMP(1);
    f32x4 acc[NT];
    float dsum[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; dsum[t] = 0.f; }
    const int nchunks = (B + 15) / 16;
    const float* Xc = b.xhat_xs + ft * 16 + li;
    for (int c0 = wave + NW * slab; c0 < nchunks; c0 += NW * CB * nslab) {
        int rr[CB][4];                                                // row indices fit 32 bits (isk_set_data checks N)
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
                // the delta rows first: the sample addresses depend on the row indices fetched above (in-order issue)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* drow = b.delta1 + (long long)(r0 + j) * H1p;
#pragma unroll
                    for (int q = 0; q < NB4; ++q) bq[j][q] = *(const f32x4*)(drow + 64 * q + 4 * li);
#pragma unroll
                    for (int u = 0; u < NR; ++u) br[j][u] = drow[64 * NB4 + 16 * u + li];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) av[j] = Xc[(long long)rr[cb][j] * b.Fp];
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
    if (slab == 0) {
        for (long long e = red_grp; e < PS; e += red_groups) {
            if (e >= 1 && e < 1 + H1) continue;     // D is rebuilt below
            float racc = 0.f;
            int w0 = li;
            if (e == red_grp && red_pre) { racc = (red_pv[0] + red_pv[1]) + (red_pv[2] + red_pv[3]); w0 += 64; }
            for (; w0 < nparts; w0 += 64) {          // 4 independent loads per lane in flight
                float pv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int w = w0 + 16 * u; pv[u] = (w < nparts) ? b.part[(long long)w * PS + e] : 0.f; }
                racc += (pv[0] + pv[1]) + (pv[2] + pv[3]);
            }
#pragma unroll
            for (int sft = 8; sft > 0; sft >>= 1) racc += __shfl_xor(racc, sft, WAVE);
            const long long gi = part_to_grad(s, e);
            if (OPT) {
                // (the launcher guarantees one element per group: e == red_grp, operands prefetched.)  The tail parameter behind
                // this element is updated by lane 0 of the group once the batch loss is known, below
                sp_racc = racc;
                if (e == 0 && li == 0) { gout[s.nparams] = racc; gout[s.nparams + 1] = o_peer; }
            } else {
                if (li == 0 && gi >= 0) gout[gi] = racc;
                if (e == 0 && li == 0) gout[s.nparams + 1] = b.rank_err ? (float)*b.rank_err : 0.f;     // behind the loss: this rank's error flag
            }
        }
    }
    GSTAMP(2);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        *(f32x4*)(red + ((wave * NT + t) * 64 + lane) * 4) = acc[t];
        float d = dsum[t];
        d += __shfl_xor(d, 16, WAVE);
        d += __shfl_xor(d, 32, WAVE);
        if (lq == 0) Dred[(wave * NT + t) * 16 + li] = d;
    }
    // what the epilogue needs from memory is requested here (the accumulators are parked, registers are free) and arrives
    // under the two barriers: this thread's share of W1 (fragment-linear: contiguous) and its feature's LayerNorm scale / bias
    float w1v[NE];
    if (OPT) {
        // every operand of the optimizer step this thread will take: requested HERE, where the accumulators are parked and registers
        // are free -- the loads arrive under the two barriers of the reduction (held across the main loop they cost 264 bytes of
        // scratch per lane)
        if (ln && tid < 32) {
            const long long gi = ((tid >> 4) ? s.off_lnb : s.off_lns) + ft * 16 + (tid & 15);
            ln_th = oa.theta[gi]; ln_m = oa.m[gi]; ln_v = adam ? oa.v[gi] : 0.f;
        }
        if (ft == 0 && tid < H1) {
            const long long gi = s.off_b[0] + tid;
            b1_th = oa.theta[gi]; b1_m = oa.m[gi]; b1_v = adam ? oa.v[gi] : 0.f;
        }
        if (red_pre && red_grp != 0) {
            sp_gi = part_to_grad(s, red_grp);
            if (sp_gi >= 0) { sp_th = oa.theta[sp_gi]; sp_m = oa.m[sp_gi]; sp_v = adam ? oa.v[sp_gi] : 0.f; }
        }
        if (tid < 64) lossp = tid < nparts ? b.part[(long long)tid * PS] : 0.f;      // (one row slab: nparts <= 48 units)
        o_coll = oa.loss_acc[1];
        o_peer = b.rank_err ? (float)*b.rank_err : 0.f;
        bc1 = oa.bt ? (float)(1.0 - oa.bt[0]) : oa.bc1;
        bc2 = oa.bt ? (float)(1.0 - oa.bt[1]) : oa.bc2;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + NTH * i;
            const long long gi = wbase + (e < NT * 256 ? e : 0);
            o_th[i] = oa.theta[gi]; o_m[i] = oa.m[gi]; o_v[i] = adam ? oa.v[gi] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) w1v[i] = OPT ? ((ln && tid + NTH * i < NT * 256) ? o_th[i] : 0.f) : ((ln && tid + NTH * i < NT * 256) ? W1n[tid + NTH * i] : 0.f);

    const float gk = ln ? b.theta[s.off_lns + k] : 1.f, bk = ln ? b.theta[s.off_lnb + k] : 0.f;   // LayerNorm blocks are padded to Fp
    __syncthreads();
    if (OPT && tid < 64) {
        // the batch loss, summed the same way by every workgroup (the same skip / apply decision everywhere); behind the first
        // barrier: its load has had the barrier's wait to arrive
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lossp += __shfl_xor(lossp, o, WAVE);
        if (tid == 0) psh[NW * 8] = lossp;
    }
    if (tid < NT * 16) {                                               // D[o], o = 16 t + li = tid
        float d = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) d += Dred[w * NT * 16 + tid];
        Dsh[tid] = d;
        if (!OPT && ft == 0 && tid < H1) gout[s.off_b[0] + tid] = d;    // db1 == D (this slab's rows)
        if (OPT) b1_grad = d;
    }
    __syncthreads();
    GSTAMP(3);
    // OPT: the step's verdict (opt_step_kernel's): a non-finite batch loss -- or an earlier one in this epoch -- skips every update
    const float lstep = OPT ? psh[NW * 8] : 0.f;
    const bool bad = OPT && ((o_coll != 0.0) || !isfinite(lstep) || o_peer != 0.f);
    if (OPT && ft == 0 && tid == 0) {
        double* const loss_acc = oa.loss_acc;
        double acc0 = loss_acc[0];
        if (o_peer != 0.f) loss_acc[3] = (double)o_peer;
        if (bad) loss_acc[1] = 1.0;
        else { acc0 += (double)lstep; loss_acc[0] = acc0; }
        if (oa.host_pub) {
            oa.host_pub[0] = acc0;
            oa.host_pub[1] = bad ? 1.0 : 0.0;
            ((int*)(oa.host_pub + 2))[0] = ((const int*)(loss_acc + 2))[0];
            oa.host_pub[3] = o_peer != 0.f ? (double)o_peer : loss_acc[3];
            __threadfence_system();
        }
    }
    if (OPT && !bad) {
        if (ft == 0 && tid < H1) {                                      // layer-1 bias: gradient D
            opt_rule(oa, bc1, bc2, b1_grad, b1_th, b1_m, b1_v);
            const long long gi = s.off_b[0] + tid;
            oa.theta[gi] = b1_th; oa.m[gi] = b1_m; if (adam) oa.v[gi] = b1_v;
        }
        if (sp_gi >= 0 && li == 0 && red_pre) {                         // this group's tail parameter
            opt_rule(oa, bc1, bc2, sp_racc, sp_th, sp_m, sp_v);
            oa.theta[sp_gi] = sp_th; oa.m[sp_gi] = sp_m; if (adam) oa.v[sp_gi] = sp_v;
        }
    }
    // Epilogue.  Thread tid keeps accumulator slot (lane' = (tid/4) % 64, r = tid % 4) for the tiles t = tid/256 + (NTH/256) i:
    // feature k = 16 ft + 4 (lane'/16) + r, output 16 t + lane'%16.  Element e = 256 t + 4 lane' + r of the workgroup's
    // contiguous share of the (fragment-linear) layer-1 gradient.
    float* gW1 = gout + s.off_W[0] + (long long)ft * NT * 256;
    float pg = 0.f, pb = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = tid + NTH * i;
        if (e < NT * 256) {
            float G = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) G += red[w * NT * 256 + e];
            float gW = G;
            if (ln) {
                const float Do = Dsh[(e >> 8) * 16 + lip];
                gW = fmaf(gk, G, __fmul_rn(bk, Do));        // (pinned: the same bits whether stored or consumed by the optimizer rule)
                pg += w1v[i] * G;
                pb += w1v[i] * Do;
            }
            if (!OPT) gW1[e] = gW;
            else if (!bad) {
                float th = o_th[i];
                opt_rule(oa, bc1, bc2, gW, th, o_m[i], o_v[i]);
                oa.theta[wbase + e] = th; oa.m[wbase + e] = o_m[i];
                if (adam) oa.v[wbase + e] = o_v[i];
            }
        }
    }
    if (ln) {
        // dg[k], dbeta[k]: sum over the 16 output columns (lane bits 2..5 of this wave) and over the tiles (the waves
        // with the same wave % 4 hold the same features)
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) { pg += __shfl_xor(pg, o, WAVE); pb += __shfl_xor(pb, o, WAVE); }
        if (lane < 4) { psh[(wave * 2 + 0) * 4 + lane] = pg; psh[(wave * 2 + 1) * 4 + lane] = pb; }
        __syncthreads();
        if (tid < 32) {
            const int which = tid >> 4, kl = tid & 15, q = kl >> 2, r = kl & 3;     // feature 16 ft + kl: lq' = q, r
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w)
                if ((w & 3) == q) v += psh[(w * 2 + which) * 4 + r];
            if (!OPT) gout[(which ? s.off_lnb : s.off_lns) + ft * 16 + kl] = v;
            else if (!bad) {
                opt_rule(oa, bc1, bc2, v, ln_th, ln_m, ln_v);
                const long long gi = (which ? s.off_lnb : s.off_lns) + ft * 16 + kl;
                oa.theta[gi] = ln_th; oa.m[gi] = ln_m; if (adam) oa.v[gi] = ln_v;
            }
        }
    }
    GSTAMP(4);
}

template <int NT, int NW>
static void launch_grad1n(const TrainShape& s, const TrainBuffers& b, const int* perm, int B, int nslab, const OptArgs* fuse, hipStream_t st) {
    const int nparts = (B + 15) / 16;
    const size_t lds = (size_t)(NW * NT * 256 + NW * NT * 16 + NT * 16 + NW * 8 + 4) * sizeof(float);
    const OptArgs none = {};
    if (fuse) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)train_grad1n_kernel<NT, NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((train_grad1n_kernel<NT, NW, true>), dim3(b.Fp / 16, 1), dim3(NW * 64), lds, st, s, b, perm, B, nparts, *fuse);
        return;
    }
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)train_grad1n_kernel<NT, NW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((train_grad1n_kernel<NT, NW, false>), dim3(b.Fp / 16, nslab), dim3(NW * 64), lds, st, s, b, perm, B, nparts, none);
}

// the optimizer step can ride on train_grad1n: one row slab, at most 64 units (one wave sums their loss partials), and every
// element of the small-gradient partial vector has a 16-lane group of its own (its optimizer operands are prefetched per group)
bool isk_train_grad1n_can_fuse(const TrainShape& s, int Fp, int B, int nslab, int nw8) {
    const int nt1 = isk_pad16(s.widths[1]) / 16;
    const int NW = (nw8 || nt1 > 9) ? 8 : 16;
    return nslab == 1 && (B + 15) / 16 <= 64 && part_size(s) <= (long long)(Fp / 16) * NW * 4;
}

void isk_launch_train_grad1n(const TrainShape& s, const TrainBuffers& b, const int* perm, int B, int nslab_flags, hipStream_t st, const OptArgs* fuse) {
    const int nt1 = isk_pad16(s.widths[1]) / 16;
    const int nslab = nslab_flags & 0xff;
    const int nw8 = (nslab_flags & 0x100) != 0;            // the engine decides (option grad1_nw8, or: the featurizer shares the CUs)
#define G1CASE(NT_) case NT_: if (nw8) launch_grad1n<NT_, 8>(s, b, perm, B, nslab, fuse, st); else launch_grad1n<NT_, 16>(s, b, perm, B, nslab, fuse, st); break;
    switch (nt1) {
        G1CASE(1) G1CASE(2) G1CASE(3) G1CASE(4) G1CASE(5) G1CASE(6) G1CASE(7) G1CASE(8) G1CASE(9)
        default: launch_grad1n<10, 8>(s, b, perm, B, nslab, fuse, st); break;
    }
#undef G1CASE
}

// ------------------------------------------------------------------ opt_step ---------
// Optimisers.jl 0.4.7: OptimiserChain(WeightDecay(reg), Adam | Nesterov), one thread per parameter.
// The batch loss sits in grad[n]; a non-finite loss (or an earlier one in this epoch) skips the
// update, as the reference throws before update! (src/iso.jl:186-189).
static_assert(sizeof(OptArgs) <= 128, "OptArgs: two kernarg cache lines");
template <bool PEERS>
__device__ __forceinline__ void opt_step_body(const OptArgs& a, const OptPeers& pr) {
    double* const loss_acc = a.loss_acc;
    ISK_LATENCY_CHAIN_PRIO();
    KSTAMP(4);
    const float bc1 = a.bt ? (float)(1.0 - a.bt[0]) : a.bc1, bc2 = a.bt ? (float)(1.0 - a.bt[1]) : a.bc2;
    // PEERS: element i of the step's gradient is the sum over the replicas' buffers, in rank order (loss and flag slots alike)
    auto gsum = [&](long long i) -> float {
        if (!PEERS) return a.grad[i];
        float s0 = pr.g[0][i];
        for (int r = 1; r < pr.n; ++r) s0 += pr.g[r][i];
        return s0;
    };
    const float lstep = gsum(a.n);
    const float peer = gsum(a.n + 1);                    // data-parallel: number of ranks that flagged an error in this step (else 0)
    const double collapsed = loss_acc[1];
    // this thread's (first) parameter: its operands are requested before the batch loss is looked at -- waiting for the loss
    // first put a second memory round trip in front of every update
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool in0 = i0 < a.n;
    const long long ic = in0 ? i0 : 0;
    const float th0 = a.theta[ic];
    float gs0 = gsum(ic);
    if (!PEERS && ic < a.slab_prefix)                    // [LN | W1 | b1] gradients arrive split over row slabs
        for (int sl = 1; sl < a.nslab; ++sl) gs0 += a.grad[(long long)sl * (a.n + 2) + ic];
    const float m0 = a.m[ic];
    const float v0 = a.kind == ISK_OPT_ADAM ? a.v[ic] : 0.f;
    const bool bad = (collapsed != 0.0) || !isfinite(lstep) || peer != 0.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double acc0 = loss_acc[0];
        if (peer != 0.f) loss_acc[3] = (double)peer;
        if (bad) loss_acc[1] = 1.0;
        else { acc0 += (double)lstep; loss_acc[0] = acc0; }
        if (a.host_pub) {
            // the epoch's result goes to the host as three stores into mapped pinned memory (every step publishes, the last
            // one counts): no copy kernels behind the epoch, the host only waits for an event
            a.host_pub[0] = acc0;
            a.host_pub[1] = bad ? 1.0 : 0.0;
            ((int*)(a.host_pub + 2))[0] = ((const int*)(loss_acc + 2))[0];
            a.host_pub[3] = peer != 0.f ? (double)peer : loss_acc[3];                   // sticky within the epoch
            __threadfence_system();                      // the three stores are visible to the host before the kernel's end-of-grid signal
        }
    }
    if (bad) return;
    auto update = [&](long long i, float th, float gs, float mi, float vi) {
        opt_rule(a, bc1, bc2, gs, th, mi, vi);
        a.m[i] = mi;
        if (a.kind == ISK_OPT_ADAM) a.v[i] = vi;
        a.theta[i] = th;
    };
    if (in0) update(i0, th0, gs0, m0, v0);
    for (long long i = i0 + (long long)gridDim.x * 256; i < a.n; i += (long long)gridDim.x * 256) {
        float gs = gsum(i);
        if (!PEERS && i < a.slab_prefix)                 // [LN | W1 | b1] gradients arrive split over row slabs
            for (int sl = 1; sl < a.nslab; ++sl) gs += a.grad[(long long)sl * (a.n + 2) + i];
        update(i, a.theta[i], gs, a.m[i], a.kind == ISK_OPT_ADAM ? a.v[i] : 0.f);
    }
}

__global__ __launch_bounds__(256) void opt_step_kernel(OptArgs a) {
    const OptPeers none = {};
    opt_step_body<false>(a, none);
}
__global__ __launch_bounds__(256) void opt_step_peers_kernel(OptArgs a, OptPeers p) { opt_step_body<true>(a, p); }

void isk_launch_opt_step(const OptArgs& a, hipStream_t st) {
    long long b = (a.n + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(opt_step_kernel, dim3((unsigned)b), dim3(256), 0, st, a);
}

void isk_launch_opt_step_peers(const OptArgs& a, const OptPeers& p, hipStream_t st) {
    long long b = (a.n + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    hipLaunchKernelGGL(opt_step_peers_kernel, dim3((unsigned)b), dim3(256), 0, st, a, p);
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
    for (int o = tid; o < a.H1p; o += 256) dl[o] = a.delta1[n * a.H1p + (a.native ? isk_delta1_pos(o, a.H1p) : o)];
    if (a.pairs)
        for (int i = tid; i < a.D; i += 256) gc[i] = 0.f;
    __syncthreads();
    const float* W1 = a.theta + a.off_W0;
    const float* g = ln ? a.theta + a.off_lns : nullptr;
    const float* xh = a.xhat + n * a.Fp;
    float s1 = 0.f, s2 = 0.f;
    const int nt = a.H1p / 16;
    for (int f = tid; f < a.F; f += 256) {
        // W1[o][f]: Flux layout (o fastest) on the layered path, fragment-linear on the fused path (a.native)
        const float* w = a.native ? W1 + ((long long)(f >> 4) * nt * 64 + ((f & 15) >> 2) * 16) * 4 + (f & 3) : W1 + (long long)f * a.H1;
        float acc0 = 0.f, acc1 = 0.f;
        if (a.native) {
            for (int o = 0; o < a.H1; ++o) {
                const float wv = w[(o >> 4) * 256 + (o & 15) * 4];
                if (o & 1) acc1 += dl[o] * wv; else acc0 += dl[o] * wv;
            }
        } else {
            int o = 0;
            for (; o + 1 < a.H1; o += 2) { acc0 += dl[o] * w[o]; acc1 += dl[o + 1] * w[o + 1]; }
            if (o < a.H1) acc0 += dl[o] * w[o];
        }
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
