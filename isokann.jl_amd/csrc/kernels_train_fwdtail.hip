// train_fwd1n and the narrow tail step as ONE launch (option fuse_tail; two launches per optimizer step together with the
// gradient kernel that carries the optimizer rule, kernels_train.hip).
//
// The grid is train_fwd1n's: (32-row blocks, tile groups x K slices).  Every workgroup stores its K-slice partial of z1
// write-through, waits for the stores, and one lane adds 1 to the row block's arrival counter (agent scope).  The workgroup whose
// add returns the last ticket is the only one that knows every partial of its 32 rows is in memory: it stays and runs the tail
// step of those two 16-row units (waves 0..3 and 4..7: train_tailn.hpp's unit, its partial loads sc1), everyone else leaves.
// Nobody waits for anybody -- a launch cannot hang on a missing arrival -- and the hand-off is the one MI355X_MICROARCH.md
// measures ("ONE lane of each storing workgroup ... adds to ONE unsharded counter, the workgroup whose add came last, told by the
// value its add returned"): sc1 stores, s_waitcnt vmcnt(0) in every storing wave, a workgroup barrier, the add; the adding wave
// loads behind the returned value, the other waves behind a barrier it then joins.  The last workgroup puts the counter back to 0
// (the next launch is a kernel boundary away).
//
// Measured (MI355X, C3's network; bit-identical to the two launches in every test): SLOWER -- 35.6 against 33.3 us per optimizer step
// at minibatch 100, 0.826 against 0.790 ms per iteration with cached features (10 steps of 1000 rows), 0.488 against 0.462 ms for
// one rank's share of C4.  The seam is three dependent memory-side round trips (write-through acknowledge, the ticket, the sc1
// partial loads) in front of a tail that the separate launch starts with its parameter loads already under way; a launch boundary
// costs less here.  Option fuse_tail, default 0; kept because it is the measurement the "two launches per step" question needed.
//
// Only the pairnet default activations (SIG) are instantiated: their unit fits the 128 registers train_fwd1n is held to (two
// workgroups per CU beside a featurizer wave); every other network keeps the two launches.
#include "isk_internal.h"
#ifndef ISK_EXPERIMENTAL
// shipped library: the variant is not built (it measured slower); option fuse_tail does not exist there and these are never reached
bool isk_train_fwdtail_ok(const TrainShape&) { return false; }
void isk_launch_train_fwdtail(const TrainShape&, const TrainBuffers&, const int*, int, float, unsigned*, hipStream_t) {}
int isk_train_fwdtail_counters(long long) { return 32; }            // (one line's worth: the engine allocates and zeroes it, nothing reads it)
#else
#include "train_fwd1n.hpp"
#include "train_tailn.hpp"

#define FT_CNT_STRIDE 32            /* one 128-byte line per row block's counter */

template <int NTB, bool LN, int L>
__global__ __launch_bounds__(FW1N_WAVES * 64, 4) void train_fwdtail_kernel(Fwd1nArgs a, TailNDesc d, TrainBuffers b, float inv_div, unsigned* cnt) {
    ISK_LATENCY_CHAIN_PRIO();
    extern __shared__ __attribute__((aligned(16))) float red[];
    __shared__ unsigned s_last;
    fwd1n_body<NTB, LN, true>(a, red);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // this wave's partial stores have left for memory
    __syncthreads();
    const int rb = blockIdx.x;
    if (threadIdx.x == 0) {
        const unsigned total = (unsigned)(a.ntg * a.KS);
        const unsigned old = __hip_atomic_fetch_add(cnt + rb * FT_CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = old == total - 1 ? 1u : 0u;
        if (last) __hip_atomic_store(cnt + rb * FT_CNT_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    // ---- the tail step of units 2 rb and 2 rb + 1 (a batch with an odd unit count: both halves run its last unit and store the
    // same bits -- the unit's two barriers are workgroup barriers)
    const int tid = threadIdx.x, half = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int wv = __builtin_amdgcn_readfirstlane((tid >> 6) & 3), ln = tid & 63;
    const int units = (a.B + 15) / 16;
    int unit = 2 * rb + half;
    if (unit > units - 1) unit = units - 1;
    // LDS (over train_fwd1n's reduction buffer, which every thread has left: the barriers above): per half
    //   s_zred [4][256] | s_tile [4][16 * TN_TS] | s_d1 [16][192]
    constexpr int HALF_FLOATS = TN_WAVES * 256 + TN_WAVES * 16 * TN_TS + 16 * 192;
    float* const base = red + half * HALF_FLOATS;
    float* const s_zred = base;
    float* const s_tile = base + TN_WAVES * 256;
    float* const s_d1 = s_tile + TN_WAVES * 16 * TN_TS;
    const int row_g = unit * 16 + (ln & 15);
    const int rowc = row_g < a.B ? row_g : a.B - 1;
    const int prow = a.perm ? a.perm[rowc] : rowc;
    const bool bt_owner = b.bt && rb == 0 && tid == 0;
    double bt0 = 0.0, bt1 = 0.0;
    if (bt_owner) { bt0 = b.bt[0]; bt1 = b.bt[1]; }
    TailNIO io;
    io.theta = b.theta; io.target = b.target; io.lossw = b.lossw; io.cot_mode = b.cot_mode;
    io.a1 = b.a1; io.zstride = b.zstride; io.KS = b.KS; io.z1img = nullptr; io.wlds = nullptr;
    io.delta1 = b.delta1; io.part_out = b.part + (long long)unit * d.PS; io.inv_div = inv_div;
    (void)tailn_unit<L, true, TN_WAVES, false, true>(d, io, unit, a.B, prow, s_zred, s_tile, s_d1, wv, ln);
    if (bt_owner) {
        b.bt[2] = bt0; b.bt[3] = bt1;
        b.bt[0] = bt0 * b.beta1; b.bt[1] = bt1 * b.beta2;
    }
}

// the shapes the fused launch covers: a narrow tail with the pairnet default activations
bool isk_train_fwdtail_ok(const TrainShape& s) {
    return isk_train_tail_narrow_ok(s) && s.act == ISK_ACT_SIGMOID && s.lastact == ISK_ACT_IDENTITY;
}
int isk_train_fwdtail_counters(long long maxB) { return (int)((maxB + 31) / 32) * FT_CNT_STRIDE; }

void isk_launch_train_fwdtail(const TrainShape& s, const TrainBuffers& b, const int* perm, int B, float inv_div, unsigned* cnt, hipStream_t st) {
    Fwd1nArgs a;
    a.X = b.xhat_xs; a.Fp = b.Fp; a.nch = b.Fp / 16; a.perm = perm; a.B = B;
    a.W1n = b.theta + s.off_W[0];
    a.g = s.layernorm ? b.theta + s.off_lns : nullptr;
    a.beta = s.layernorm ? b.theta + s.off_lnb : nullptr;
    a.nt = isk_pad16(s.widths[1]) / 16;
    a.ntg = (a.nt + FW1N_NTB - 1) / FW1N_NTB;
    isk_fwd1n_plan(B, b.Fp, a.nt, &a.KS, &a.cpb);
    a.zp = b.a1; a.zstride = b.zstride;
    const TailNDesc d = isk_tailn_desc(s);
    const int L = s.nlayers - 1;
    const int RB = (B + 31) / 32;
    const int ntb = (a.nt + a.ntg - 1) / a.ntg;
    const dim3 grid(RB, a.ntg * a.KS), blk(FW1N_WAVES * 64);
    constexpr size_t tail_lds = (size_t)2 * (TN_WAVES * 256 + TN_WAVES * 16 * TN_TS + 16 * 192) * sizeof(float);
#define FT_LAUNCH(N_, LN_, L_)                                                                                                   \
    {                                                                                                                            \
        size_t lds = (size_t)4 * 2 * N_ * 256 * sizeof(float);                                                                   \
        if (lds < tail_lds) lds = tail_lds;                                                                                      \
        hipLaunchKernelGGL((train_fwdtail_kernel<N_, LN_, L_>), grid, blk, lds, st, a, d, b, inv_div, cnt);                      \
    }
#define FT_L(N_, LN_)                                                                                                            \
    switch (L) { case 1: FT_LAUNCH(N_, LN_, 1) break; case 2: FT_LAUNCH(N_, LN_, 2) break; default: FT_LAUNCH(N_, LN_, 3) break; }
#define FT_CASE(N_) case N_: if (a.g) { FT_L(N_, true) } else { FT_L(N_, false) } break;
    switch (ntb) { FT_CASE(1) FT_CASE(2) FT_CASE(3) FT_CASE(4) default: if (a.g) { FT_L(5, true) } else { FT_L(5, false) } break; }
#undef FT_CASE
#undef FT_L
#undef FT_LAUNCH
}
#endif  // ISK_EXPERIMENTAL
