// train_tail for narrow tails as a kernel of its own (one launch per optimizer step between train_fwd1n and train_grad1n);
// the device code is train_tailn.hpp.
#include "train_tailn.hpp"

template <int L, bool SIG>
__global__ __launch_bounds__(TN_WAVES * 64) void train_tailn_kernel(TailNDesc d, TrainBuffers b, const int* __restrict__ perm, int B,
                                                                   float inv_div) {
    ISK_LATENCY_CHAIN_PRIO();
    __shared__ __attribute__((aligned(16))) float s_zred[TN_WAVES * 256];
    __shared__ __attribute__((aligned(16))) float s_tile[TN_WAVES * 16 * TN_TS];
    __shared__ __attribute__((aligned(16))) float s_d1[16 * 192];                         // delta1 image [16][H1p <= 192]
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
    const int unit = blockIdx.x;
    const int row_g = unit * 16 + (ln & 15);
    const int rowc = row_g < B ? row_g : B - 1;
    const int prow = perm ? perm[rowc] : rowc;
    const bool bt_owner = b.bt && unit == 0 && tid == 0;
    double bt0 = 0.0, bt1 = 0.0;
    if (bt_owner) { bt0 = b.bt[0]; bt1 = b.bt[1]; }
    TailNIO io;
    io.theta = b.theta; io.target = b.target; io.lossw = b.lossw; io.cot_mode = b.cot_mode;
    io.a1 = b.a1; io.zstride = b.zstride; io.KS = b.KS; io.z1img = nullptr;
    io.delta1 = b.delta1; io.part_out = b.part + (long long)unit * d.PS; io.inv_div = inv_div;
    (void)tailn_unit<L, SIG, TN_WAVES, false>(d, io, unit, B, prow, s_zred, s_tile, s_d1, wv, ln);
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

TailNDesc isk_tailn_desc(const TrainShape& s) {
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
    return d;
}

void isk_launch_train_tail_narrow(const TrainShape& s, const TrainBuffers& b, const int* perm, int B, float inv_div, hipStream_t st) {
    const TailNDesc d = isk_tailn_desc(s);
    const int L = s.nlayers - 1;
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
