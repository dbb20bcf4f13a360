// Internal declarations shared by the HIP kernels and the host engine.
// gfx950 (MI355X) only: wave64, v_mfma_f32_16x16x4_f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/isokann_hip.h"

#define ISK_TILE 16          // MFMA 16x16x4 tile edge; every width is padded to a multiple
#define ISK_MAX_NT 10        // layer-1 output tiles the fused forward kernel keeps in registers (H1 <= 160)
#define ISK_NUM_CU 256

static inline int isk_pad16(int x) { return (x + 15) & ~15; }
static inline int isk_pad32(int x) { return (x + 31) & ~31; }   // feature rows: the bf16 MFMA contracts 32 at a time

// ---------------------------------------------------------------- engine options ---
// Every knob that selects a kernel variant or the schedule of an iteration, PER ENGINE: isk_create reads the defaults
// (the environment variable ISK_<KEY> of the creating process overrides a default at that moment -- nothing is latched
// in a static), isk_set_option / isk_get_option change and read them afterwards.  Keys: include/isokann_hip.h.
struct IskOptions {
    int bf16_skew = 1;            // split-bf16 forward: second wave of a SIMD converts samples in the first half of a stage
    int overlap_featurize = 2;    // next iteration's featurization: 2 side stream beside training, 1 beside the forward, 0 inline
    int side_min_steps = 4;       // fewest optimizer steps of the coming training phase for the side-stream featurizer
    int side_early = 1;           // side launch right behind the forward (0: after the target chain)
    int feat_early_pct = 0;       // share of the next iteration's ys rows featurized beside THIS forward (measured: a net loss)
    int feat_early_wgs = 128;
    int feat_wg_per_cu = 1;       // throttle of the background featurizer: workgroups per CU
    int feat_bg_wgs = 0;          //                                        ... or in total (0 = per-CU rule)
    int feat_pf = 1;              // samples in flight per wave of the background featurizer
    int dense_groups = 1;         // layered bf16 forward: one grouped launch per layer
    int train_graph = 1;          // replay the training epoch as one hipGraph
    int grad1_nw8 = -1;           // train_grad1n variant: 1 = 8 waves, 0 = 16 waves, -1 = 8 while the featurizer shares the CUs
    int fwd_waves_per_simd = 1;   // residency of the f32 forward kernel
    int dp_peer_reduce = 0;       // single-process multi-GPU group (isk_dp_run_all): 1 = no gradient all-reduce at all -- every replica's
                                  // optimizer step reads the other replicas' gradient buffers directly (peer-mapped) and sums them
    int dp_bucket = 1;            // native data-parallel run, wide networks: all-reduce the gradient in two buckets, the layers >= 1 under
                                  // the first layer's backward (0: one all-reduce behind the whole backward)
    int dp_force_coll = 0;        // issue the native collectives on a one-rank communicator (tests)
    int wide_fused = 1;           // layered path, plain-bf16 forward: the fused wide kernel (0: layer by layer through HBM)
    int train_persist = 0;        // fused path, single GPU: a training epoch as ONE persistent launch (experimental: measured SLOWER
                                  // than the four launches per optimizer step it replaces -- DESIGN.md section 6c; default off)
    int fuse_opt = 1;             // training on one GPU, one row slab (batches <= 768 rows): the optimizer step rides on train_grad1n
                                  // (three launches per step instead of four)
    int fuse_opt_rows = 768;      // ... for batches of up to this many rows (<= 1024).  Beyond 768 the gradient kernel splits the rows over two
                                  // workgroups per feature tile and opt_step sums the two slabs; keeping C3's 1000-row batches on one slab
                                  // to fuse the step was measured SLOWER (1.177 against 1.129 ms per iteration: twice the MFMA chain per wave)
    int fuse_head = 1;            // layered training GEMMs, >= 3 layers, nout <= 4: last layer forward + loss + both deltas as ONE launch
    int fuse_boundary = 1;        // layered training GEMMs, no leading LayerNorm: opt_step + the next step's weight and batch images as ONE launch
    int epoch_gather = 0;         // layered training GEMMs: the batch images of a whole epoch are gathered in one launch in front of its steps.
                                  // Bit-identical, and no faster (C5 shard 2.99 against 2.95 ms: the step keeps a launch for the weight images)
    int g16_waves = 0;            // layered training GEMMs: waves per workgroup that split the contraction (4 | 8; 0: eight where the product is
                                  // at least 16 steps of 32 deep)
    int fuse_tail = 0;            // training, narrow tail with the pairnet default activations: train_fwd1n and the tail step as ONE launch
                                  // (the workgroup whose K-slice partial arrives last runs the tail of its rows: kernels_train_fwdtail.hip).
                                  // Bit-identical to the two launches and measured SLOWER (+2.3 us per step at minibatch 100): default off
    int tail_narrow = 1;          // fused path: register-resident train_tail for narrow tails (0: the generic LDS kernel)
    int train_bf16 = 0;           // layered path: training GEMMs with bf16 operands (f32 master weights, f32 accumulate): the C5
                                  // arithmetic of BASELINE.json; 0 = f32
    int debug_inject_rank_error = 0;   // tests: in the native data-parallel run this rank flags an error at its n-th optimizer step
    int debug_feat_copy = 0;      // diagnostics
    int debug_stamps = 0;
    int debug_rowmod = 0;
};

// ---------------------------------------------------------------- featurize -------
// coords/features rows -> (optionally LayerNorm-standardised) zero-padded feature rows
struct FeatArgs {
    const float* in;        // [S][D]   (D = 3*n_atoms for pdist, = F for identity)
    const int2*  pairs;     // [F] 0-based atom pairs (pdist) or nullptr (identity)
    const unsigned* packed_pairs;  // [F] a | b << 16 when n_atoms <= 65535, else nullptr
    float* xhat;            // [S][Fp]
    long long S;
    int D, F, Fp;
    int normalize;          // 1: (x-mu)*rstd, Flux normalise over the feature axis
    int ln_mode;            // isk_ln_mode
    float eps;
    int background;         // launched on the side stream next to other kernels: may be throttled
    // throttle of a background launch (engine options feat_wg_per_cu / feat_bg_wgs / feat_pf, see IskOptions); 0 = default
    int bg_wg_per_cu, bg_wgs_total, bg_pf;
    int dbg_copy;           // diagnostic (option debug_feat_copy): n x 256 workgroups of a plain byte stream instead of a background launch
    float* raw;             // optional [S][Fp]: un-normalised features (needed by the coordinate pullback)
    float* stats;           // optional [S][2]: mean, 1/denominator of the LayerNorm standardisation
    // split-f16 output (forward_mode 3; register featurizer only, see isk_featurize_reg_ok): xhat rows are written as
    // [Fp/32 chunks][32 x f16 hi | 32 x f16 lo] of xhat * 2^s (same 4*Fp bytes per row), 2^-s goes to xscale[row]
    int out_bf16;           // xhat rows are written as bf16 [S][ld16] (Fp live elements per row): the A operand of the fused wide forward
    int ld16;               //   row stride of those rows in elements (isk_ld16(Fp): whole 128-byte lines; 0 = Fp)
                            // (forward_mode 2 on the layered path); register featurizer only; not together with `planes`
    int planes;
    long long row0;         // planes only: `xhat` is the BASE of the plane buffer and local row s is global row row0 + s of its
                            // unit-blocked layout (isk_plane_off): a unit = 16 consecutive rows, per 32-feature chunk one 2 KiB
                            // block [hi: 16 rows x 64 B | lo: 16 rows x 64 B], so that one MFMA operand load of the forward
                            // (16 rows x 64 B) is ONE contiguous KiB = 8 whole 128-byte lines instead of half of 16 lines
    float* xscale;          // [S] inverse row scales (planes only)
    float* xhat32;          // optional: a second copy, always f32, of rows >= x32_first at xhat32[(row - x32_first)][Fp] (the xs rows the
    long long x32_first;    //           training kernels read)
};
// float offset of global row g inside the unit-blocked plane buffer (chunk 0, hi plane, first feature pair); chunk c adds
// 512 floats (2 KiB), the lo plane 256 floats (1 KiB).  Rows are padded to whole units: isk_plane_rows(S).
__host__ __device__ static inline long long isk_plane_off(long long g, int Fp) { return (g >> 4) * 16 * (long long)Fp + (g & 15) * 16; }
static inline long long isk_plane_rows(long long S) { return (S + 15) / 16 * 16; }
void isk_launch_featurize(const FeatArgs& a, hipStream_t st);
bool isk_featurize_reg_ok(const FeatArgs& a);

// ---------------------------------------------------------------- forward ---------
// Weight layout "fragment-linear": for layer with Kp inputs (multiple of 16) and Np outputs
// (multiple of 16):  Wt[((c*(Np/16) + t)*64 + lane)*4 + j] = W[o = t*16 + (lane&15)][k = c*16 + 4*(lane>>4) + j]
// i.e. exactly the B operand of 4 consecutive v_mfma_f32_16x16x4_f32 for k-chunk c and output tile t,
// so one coalesced 1 KiB dwordx4 load per wave fetches it.
struct TailLayer {
    const float* Wt;        // fragment-linear, [kchunks][ntiles][64][4]
    const float* bias;      // [ntiles*16], zero padded
    int kchunks, ntiles, act, width;
    int lds_off;            // float offset of this layer's weights inside the workgroup's LDS copy
};
struct MlpFwdArgs {
    const float* X;         // [S][Fp] standardised features (row-major, zero padded)
    const int*   rows;      // optional gather: logical row r reads X[rows[r]] (nullptr = identity)
    long long S;
    int Fp, nchunks;        // nchunks = Fp/16
    const float* W1t;       // fragment-linear layer-1 weights with LayerNorm scale folded in
    const float* bias1;     // [NT*16] layer-1 bias with W1*ln_bias folded in
    int act1, H1, nt1;
    int ntail;
    TailLayer tail[ISK_MAX_LAYERS];
    float* out;             // [S][nout]
    int nout;
    // dense-block mode (layer-by-layer path for networks too wide for the fused kernel): ntail == 0 and the
    // kernel computes columns [16*w_tile0, 16*(w_tile0+nt1)) of ONE layer, writing act(z) for every padded column
    int dense_mode, w_nt_total, w_tile0, out_ld, out_col0;
    // dense_groups = G > 1: ONE launch computes G adjacent column blocks of nt1 tiles each.  Workgroups b, b+8, .. b+8(G-1)
    // (same XCD, started together) take the same rows and different column blocks, so the sample rows they stream in
    // lockstep are fetched from HBM once and re-read from that XCD's L2 (bf16 kernels only)
    int dense_groups;
    int tailw_floats;       // total floats staged in LDS per workgroup: tail weights, then the bias block
    int bias_lds0;          // float offset of the bias block: [bias1: nt1*16][tail 0 bias: ntiles*16]...
    int stride_a, stride_b; // floats per row of the two per-wave LDS ping-pong buffers (padded width + 4)
    int sched_skew;            // bf16x3 pipelined kernel: second wave of each SIMD converts early (0 = same schedule)
    int waves_per_simd;        // f32 forward kernel: residency (engine option fwd_waves_per_simd; 0 = 1)
    // split-f16 forward (forward_mode 3): X holds pre-split f16 planes (FeatArgs.planes), xscale the inverse row scales,
    // wscale[0] the inverse scale of the weight planes (isk_launch_fold_f16); z = acc * xscale[row] * wscale[0]
    const float* xscale;
    const float* wscale;
    long long x_row0;          // split-f16 only: X is the base of the unit-blocked plane buffer, logical row r is its row x_row0 + r
    unsigned long long* dbg;   // diagnostic builds only (-DISK_STAMP): per-wave phase cycle sums, else nullptr
};
// append the bias block to the LDS image (call once nt1 / ntail / tail[] / tailw_floats are final)
static inline void isk_fwd_layout_biases(MlpFwdArgs& a) {
    a.bias_lds0 = a.tailw_floats;
    int n = a.nt1 * 16;
    for (int l = 0; l < a.ntail; ++l) n += a.tail[l].ntiles * 16;
    a.tailw_floats += n;
}
int  isk_launch_mlp_fwd(const MlpFwdArgs& a, hipStream_t st);   // returns 0 or ISK_EUNSUPPORTED
// split-bf16 variant: W16 = three bf16 planes of the layer-1 weights from isk_launch_fold_bf16
// planes: 3 = split-bf16 (f32-equivalent), 1 = plain bf16 operands (forward_mode 2); the image from isk_launch_fold_bf16 must match
int  isk_launch_mlp_fwd_bf16x3(const MlpFwdArgs& a, const void* W16, int planes, hipStream_t st);
void isk_launch_fold_bf16(const float* Wt, void* out, int Kp, int nt, int planes, hipStream_t st);
// split-f16 variant (forward_mode 3): W16 = two scaled f16 planes from isk_launch_fold_f16, samples pre-split by the featurizer
int  isk_launch_mlp_fwd_f16x2(const MlpFwdArgs& a, const void* W16, hipStream_t st);
int  isk_mlp_fwd_f16x2_fits(const MlpFwdArgs& a);   // 1 when the shape's LDS footprint fits the split-f16 kernel (a.xscale / a.wscale set)
// wmax: the tensor maximum max |Wt| as float bits (built by fold_kernel), wscale_inv[0] <- inverse scale of the planes
void isk_launch_fold_f16(const float* Wt, void* out, int Kp, int nt, const unsigned* wmax, float* wscale_inv, hipStream_t st);
#define ISK_MAX_NT_BF16 16   // output tiles per workgroup of the plain-bf16 forward
int  isk_bf16_chunks(int Kp);     // 32-feature chunks in the plane image (padded to the kernel's ring size)

// ---------------------------------------------------------------- fused bf16 forward for wide networks (kernels_fwd_wide.hip) ----
// row stride (elements) of the bf16 feature rows: every row starts on a 128-byte line, so a 256-byte run of a row is two whole
// lines (at the plain stride of Fp = 2016 elements a run touched three: FETCH_SIZE was 1.33 x the algorithmic bytes)
// ... but never a multiple of 1 KiB: at a power-of-two stride (4096 B for C5) the rows of a block alias onto the same HBM channels
static inline int isk_ld16(int Fp) { int ld = (Fp + 63) & ~63; if (ld % 512 == 0) ld += 64; return ld; }
struct WideFwdArgs {
    const void* X; long long S; int Fp;           // [S][ldx] standardised feature rows (Fp live elements, zero padded to a multiple of
    int x_bf16;                                   //   32), bf16 as the featurizer writes them for this path (FeatArgs.out_bf16; must be 1)
    int ldx;                                      //   row stride in elements (>= Fp; isk_ld16(Fp): rows start on 128-byte lines)
    int nhidden;                                  // hidden layers (all layers but the last), each of width <= 512
    int N[ISK_MAX_LAYERS], Kp[ISK_MAX_LAYERS];    // width and padded contraction (multiple of 32) per hidden layer
    const void* Wt[ISK_MAX_LAYERS];               // fragment-linear bf16 images (WtImagesArgs.Wf; columns >= N zero): isk_launch_wt_images
    const float* bias[ISK_MAX_LAYERS];
    int act;
    const float* Wlast; const float* blast;       // last layer, f32, Flux layout (out, in) column-major; Klast = N[nhidden - 1]
    int Klast, nout, lastact;
    float* out; long long out_ld;                 // [S][out_ld]
};
int isk_wide_fwd_ok(const WideFwdArgs& a);
int isk_launch_wide_fwd(const WideFwdArgs& a, hipStream_t st);

// ---------------------------------------------------------------- layered (wide-network) path ----
void isk_launch_act(const float* z, float* a, long long rows, int ld, int width, int act, hipStream_t st);
// delta_L and the batch loss (written to loss_slot[0]; loss_slot[1] <- *rank_err or 0, see TrainBuffers.rank_err) from the
// last layer's pre-activations
void isk_launch_loss_delta(const float* z, int ld, const float* target, const float* lossw, const int* perm, int B, int nout,
                           float inv_div, int lastact, float* delta, int delta_ld, float* loss_slot, const int* rank_err, hipStream_t st);
// cotangent seed of the coordinate pullback on the layered path: delta_L = cot .* lastact'(z_L), zero padded rows / columns
void isk_launch_cot_delta(const float* z, int ld, const float* cot, long long B, int nout, int lastact, float* delta, int delta_ld,
                          hipStream_t st);
// G[k][o] = sum_n x[row(n)][k] * delta[n][o] written in Flux layout gW[o + k*N] (or to a temporary for the LayerNorm unfold);
// db[o] = sum_n delta[n][o]
void isk_launch_dense_dw(const float* x, int x_ld, const int* perm, const float* delta, int delta_ld, int B, int K, int N,
                         float* gW, float* gb, hipStream_t st);
// delta_prev = dz .* act'(z_prev) (dz from the transposed-weight GEMM), zero padded
void isk_launch_dact_mul(const float* dz, const float* z, float* delta_prev, long long rows, int ld, int width, int act, hipStream_t st);
// fragment-linear image of W' (for delta_prev = delta * W): WtT[(c*(Kp/16) + t)*64 + lane][j] = W[o = 16c + 4q + j][k = 16t + li]
void isk_launch_fold_T(const float* theta, long long off_W, int K, int N, float* WtT, hipStream_t st);
// LayerNorm unfold of layer 0 from the raw G (Flux layout): dW1 = g.*G + beta*D', dg = sum_o W1.*G, dbeta = W1'*D
void isk_launch_ln_unfold(const float* G, const float* D, const float* theta, long long off_W, long long off_lns, long long off_lnb,
                          int F, int H1, float* grad, hipStream_t st);

// ---------------------------------------------------------------- bf16 training GEMMs (kernels_gemm16.hip) ----
// C[i][j] = sum_k A[i][k] * B[j][k]  ("NT": both operands contiguous along the contraction), bf16 MFMA, f32 accumulate
enum { ISK_G16_FWD = 0, ISK_G16_DX = 1, ISK_G16_DW = 2 };
struct Gemm16Args {
    const void* A; long long lda; int a_f32;     // [M][lda]   f32 (converted on load) or bf16
    const void* B; long long ldb; int b_f32;     // [Nn][ldb]
    int M, Nn, Kc, kvalid;                       // Kc: contraction length, a multiple of 32 (operands zero padded); kvalid < Kc:
                                                 // elements k >= kvalid are NOT readable padding and are masked (slow path)
    int epi;
    // ISK_G16_FWD: z = C + bias (f32 [M][ldz]); images receive act(z).  ISK_G16_DX: images receive C .* act'(zprev)
    const float* bias; int act;
    float* z; int ldz;
    const float* zprev; int ldzp;
    float* o32; int ld32;                        // optional f32 image of the epilogue value (FWD)
    void* o16; int ld16, M16;                    // optional row-major bf16 image [M16][ld16]: rows >= M and columns >= Nn are zeros
    void* ot16; int ldt;                         // optional TRANSPOSED bf16 image [Nn][ldt]: columns i in [M, ldt) are zeros
    // ISK_G16_DW: gout[i * gld + j] = C (f32)
    float* gout; long long gld;
    int waves;                                   // waves that split the contraction: 4 | 8, 0 = by depth (8 from 16 steps on)
    int f32math;                                 // 1: Float32 arithmetic -- f32 fragments (a_f32 = b_f32 = 1), eight v_mfma_f32_16x16x4_f32 per
                                                 // 32-deep step instead of one bf16 MFMA -- and the images o16 / ot16 are FLOAT arrays
};
int  isk_launch_gemm16(const Gemm16Args& a, hipStream_t st);
#define ISK_G16_MAXGROUP 3
#define ISK_HEAD_ROWS 8                          /* rows per workgroup of the head kernel */
#define ISK_HEAD_MAXG 256                        /* workgroups of the head kernel whose partials the next launch sums (8 rows each) */
struct Gemm16Head {                              // side job of a grouped launch: sum the head kernel's per-workgroup partials
    const float* part; long long stride; int G;  // [G][stride]
    long long len;                               // out[0 .. len) = sums of elements 0 .. len-1 (dW_L | db_L); element len is the batch loss
    float* out; float* loss; const int* rank_err;   // loss[0] = batch loss, loss[1] = this rank's error flag (as loss_delta16 leaves them)
};
struct Gemm16Group {                             // several independent products as ONE launch (kernels_gemm16.hip)
    Gemm16Args g[ISK_G16_MAXGROUP];
    int n, gx[ISK_G16_MAXGROUP], first[ISK_G16_MAXGROUP];
    int slow[ISK_G16_MAXGROUP];                  // member takes the elementwise operand path (ragged / unaligned shapes)
    int f32[ISK_G16_MAXGROUP];                   // member runs in Float32 arithmetic
    const float* head_part; long long head_stride; int head_G; long long head_len; float* head_out; float* head_loss; const int* head_rank_err;
};
int  isk_launch_gemm16_group(const Gemm16Args* list, int n, hipStream_t st, const Gemm16Head* head = nullptr);
// the head of the backward pass as one launch (kernels_gemm16.hip head_kernel): last layer forward + loss + delta_L + delta_{L-1}
// images + per-workgroup partials of dW_L | db_L | loss (summed by the next grouped launch through Gemm16Head)
bool isk_head_ok(int K, int N, int B, int Bp, int lda, int ldzp);
void isk_launch_head(const void* a, int lda, int a_f32, const float* zprev, int ldzp, int act, const float* W, const float* bias, int K, int N,
                     const float* target, const float* lossw, const int* perm, int B, int Bp, float inv_div, int lastact, void* d16, int ld16,
                     void* dt16, long long ldt, int img_f32, float* part, long long pstride, hipStream_t st);
// gathered batch rows -> xb [Bp][Kp] bf16 and xt [K + 1][ldt] bf16 (row K = ones; columns [B, Bp) zeros)
// img_f32: the images are FLOAT arrays (Float32 arithmetic of the training GEMMs) instead of bf16
void isk_launch_gather_images(const float* X, int ldx, const int* perm, int B, int Bp, int K, int Kp, void* xb, void* xt, long long ldt,
                              int img_f32, hipStream_t st);
// ... of every batch of an epoch: batch y reads perm[y * B .. + B) and writes its images y * stride bytes further
void isk_launch_gather_epoch(const float* X, int ldx, const int* perm, int B, int Bp, int K, int Kp, void* xb, void* xt, long long ldt,
                             int img_f32, int nbatch, long long xb_stride_bytes, long long xt_stride_bytes, hipStream_t st);
struct WtImagesArgs {
    const float* theta; const float* scale0;     // scale0: LayerNorm scale folded into layer 0's image (or nullptr)
    int nlayers;
    long long off_W[ISK_MAX_LAYERS];
    int K[ISK_MAX_LAYERS], N[ISK_MAX_LAYERS], Kp[ISK_MAX_LAYERS];
    void* Wt[ISK_MAX_LAYERS];                    // [N][Kp] bf16 (nullptr: not wanted)
    void* Wf[ISK_MAX_LAYERS];                    // fragment-linear [pad128(K) / 32][32][64 lanes][8] bf16 of a layer <= 512 wide, zero
                                                 //   initialised by the owner (steps >= Kp / 32 are never written); nullptr: not wanted
    int first_block[ISK_MAX_LAYERS];             // filled by the launcher
    int wt_f32;                                  // 1: the Wt images are FLOAT arrays (Float32 arithmetic of the training GEMMs)
};
void isk_launch_wt_images(WtImagesArgs& a, hipStream_t st);
// the boundary between two optimizer steps of the wide-network path as one launch: optimizer step + the next step's weight images +
// the next step's batch images (next_perm == nullptr: none); networks without a leading LayerNorm
struct OptArgs;
void isk_launch_opt_prep(const OptArgs& oa, WtImagesArgs& w, const long long* off_b, const float* X, int ldx, const int* next_perm, int B, int Bp,
                         int K, int Kp, void* xb, void* xt, long long ldt, int img_f32, hipStream_t st);
// isk_launch_gather_images + isk_launch_wt_images as ONE launch (the head of an optimizer step of the bf16 layered path)
void isk_launch_prep_images(const float* X, int ldx, const int* perm, int B, int Bp, int K, int Kp, void* xb, void* xt, long long ldt,
                            int img_f32, WtImagesArgs& w, hipStream_t st);
void isk_launch_loss_delta16(const float* z, int ld, const float* target, const float* lossw, const int* perm, int B, int Bp, int nout,
                             float inv_div, int lastact, void* d16, int ld16, void* dt16, long long ldt, int img_f32, float* loss_slot,
                             const int* rank_err, hipStream_t st);

// parameter "fold": flat Flux-order theta -> fragment-linear weights and folded bias
struct FoldArgs {
    const float* theta;
    int nlayers;
    int widths[ISK_MAX_LAYERS + 1];
    int Kp[ISK_MAX_LAYERS];               // padded input width per layer (layer 0: Fp; others pad16, or pad32 on the layered path)
    long long off_lns, off_lnb;           // offsets in theta, -1 if no LayerNorm
    long long off_W[ISK_MAX_LAYERS], off_b[ISK_MAX_LAYERS];
    float* Wt[ISK_MAX_LAYERS];            // fragment-linear per layer
    float* bias[ISK_MAX_LAYERS];          // padded per layer (layer 0: folded)
    double* bt; double beta1, beta2;      // optional: advance the device copy of [beta1^t, beta2^t] (one thread; see OptArgs)
    int native;                           // 1: theta is in the fused path's engine layout (layer 0 fragment-linear, LayerNorm blocks padded)
    unsigned* wmax;                       // optional (native): max |W1g| of the folded layer-1 image as float bits, built with one atomic per
    unsigned* wmax_clear;                 //   workgroup; the OTHER slot is cleared for the next fold (forward_mode 3: scale of the f16 planes)
};
void isk_launch_fold(const FoldArgs& a, hipStream_t st);

// K-mean  (expectation, src/isotarget.jl:18) + min/max for shift-scale
struct KmeanArgs {
    const float* out_ys;    // [N*K][nout]
    float* ks;              // [N][nout]
    double* ks64 = nullptr; // optional [N][nout]: the same means UNROUNDED, for the Float64 simplex solve (IsaArgs.ks64)
    const float* w;         // optional [N][K] sample weights (WeightedSamples / Girsanov): sum_k w*f / K
    long long N; int K, nout;
    double* zero2;          // optional: the coming epoch's accumulator [loss sum, collapsed flag, (status word), peer flag]: elements 0, 1
                            // and 3 are zeroed by the first thread (saves a memset node)
};
void isk_launch_kmean(const KmeanArgs& a, hipStream_t st);
void isk_launch_minmax(const float* v, long long n, float* mm /*[2] device*/, hipStream_t st);
// target = (ks - min)/(max - min); status[0] |= 1 if !(max > min)
void isk_launch_shiftscale(const float* ks, const float* mm, float* target, long long n, int* status, hipStream_t st);
// target[n][a] = sum_b At[a][b]*ks[n][b] rows permuted: target[n][i] = (At ks)[rowperm[i]]
void isk_launch_isa_apply(const float* ks, const double* At /*[d*d] device, row-major At[a][b], Float64*/, const int* rowperm,
                          float* target, long long N, int d, hipStream_t st);

// device-side Float64 inner simplex algorithm + fixperm + loss weights (single workgroup)
struct IsaArgs {
    const float* ks;        // [N][d]
    const double* ks64 = nullptr;   // optional [N][d]: K chi as the K-mean summed it in double, before the rounding to Float32.  The reference
                            // rounds here (expectation returns the model's eltype, src/isotarget.jl:18,103, and myisa widens it again:
                            // :82); on villin-like data that one rounding, amplified by the simplex inverse iteration after iteration, IS
                            // the "fp32 floor" of a run against a Float64 trajectory (scripts/floor_study.py: 2.5e-5 after 10 iterations
                            // with everything else exact).  A wider intermediate, never a less accurate one.  nullptr: widen ks
    const float* chi;       // [N][d]
    double* work;           // isk_isa_work_doubles(N, d) Float64 scratch
    float* target;          // [N][d]
    float* lossw;           // [d]
    int* status;            // bit 1 (value 2): singular / non-finite
    long long N; int d, permute, whitening;
    long long out_first, out_count;   // rows [out_first, out_first + out_count) of the target are written (data-parallel: the local shard)
};
void isk_launch_isa(const IsaArgs& a, hipStream_t st);
long long isk_isa_work_doubles(long long N, int d);

// ---------------------------------------------------------------- training --------
struct TrainShape {
    int nlayers;
    int widths[ISK_MAX_LAYERS + 1];
    int act, lastact;
    int layernorm;
    long long off_lns, off_lnb;
    long long off_W[ISK_MAX_LAYERS], off_b[ISK_MAX_LAYERS];
    long long nparams;
};
// column of output o inside a delta1 row of H1p = 16 nt floats: the first 4*(nt/4) tiles are interleaved four at a time so
// that lane c's dwordx4 at 64 q + 4 c yields column c of tiles 4q .. 4q+3; the remaining tiles keep their place
__host__ __device__ static inline int isk_delta1_pos(int o, int H1p) {
    const int nb4 = (H1p / 16) / 4;
    if (o >= 64 * nb4) return o;
    const int t = o >> 4, c = o & 15;
    return 64 * (t >> 2) + 4 * c + (t & 3);
}
#define ISK_FWD1_MAXKS 8     // K slices of train_fwd1n (partials summed by train_tail)
struct TrainBuffers {
    const float* xhat_xs;   // [N][Fp]
    const float* target;    // [N][nout]
    const float* lossw;     // [nout]
    const float* theta;     // flat params, engine layout (fused path: layer-1 weights fragment-linear, see kernels_train.hip)
    const float* W1t;       // folded layer-1 weights (fragment-linear)            -- layered path only
    const float* bias1;     // folded layer-1 bias                                  -- layered path only
    float* a1;              // [KS][Bpad32][H1p]  K-slice partials of the layer-1 pre-activations (train_fwd1n -> train_tail)
    long long zstride;      // floats between two K-slice partials
    int KS;                 // K slices the coming train_tail sums (isk_fwd1n_plan of the batch size)
    double* bt;             // optional [4]: beta^t of the coming step [0..1]; train_tail hands them to opt_step in [2..3] and advances
    double beta1, beta2;
    float* delta1;          // [Bpad][H1p]     dL/dz1, columns in isk_delta1_pos order
    float* part;            // per-workgroup partial sums from the tail kernel
    float* grad;            // flat gradient [nparams] + [loss, nonfinite]
    double* loss_acc;       // [2]: epoch loss sum (double), collapsed flag (as double)
    int cot_mode;           // 1: `target` is a cotangent (coordinate pullback), see isk_launch_train_tail
    int tail_generic;       // 1: always the generic train_tail_kernel (option tail_narrow = 0); 0: the register-resident kernel where the tail is narrow
    int Fp;
    const int* rank_err;    // optional device word: this rank's error flag.  The gradient kernel copies it (as a float) into
                            // grad[nparams + 1], the slot behind the batch loss, so the data-parallel all-reduce of [grad | loss |
                            // flag] tells EVERY rank in the same step that some rank failed; nullptr writes 0
};
struct Fwd1nArgs {
    const float* X; int Fp, nch; const int* perm; int B;
    const float* W1n; const float* g; const float* beta;      // g == nullptr: no LayerNorm
    int KS, cpb;                                              // K slices, 16-feature chunks per slice
    int nt, ntg;                                              // layer-1 output tiles, tile groups (<= 5 tiles per workgroup)
    float* zp; long long zstride;
};
void isk_fwd1n_plan(int B, int Fp, int nt, int* KS, int* cpb);
// batch rows are perm[first .. first+B); fused path: K-slice partials into b.a1
void isk_launch_train_fwd1n(const TrainShape& s, const TrainBuffers& b, const int* perm, int B, hipStream_t st);
// out[n][.] (row stride 16*nt) = X[row(n)][0..ldx) * Wt + bias   (train_fwd1_kernel on any layer; ldx multiple of 16)
void isk_launch_gemm_rows(const float* X, int ldx, const float* Wt, const float* bias, int nt, const int* perm, int B, float* out,
                          hipStream_t st);
void isk_launch_train_tail(const TrainShape& s, const TrainBuffers& b, const int* perm, int B, float inv_div, hipStream_t st);
// kernels_train_tailn.hip: the same step for narrow tails (every tail layer <= 16 wide), one 16-sample unit in registers
bool isk_train_tail_narrow_ok(const TrainShape& s);
void isk_launch_train_tail_narrow(const TrainShape& s, const TrainBuffers& b, const int* perm, int B, float inv_div, hipStream_t st);
// kernels_train_fwdtail.hip: train_fwd1n + the narrow tail step as one launch; cnt: isk_train_fwdtail_counters(max batch) zeroed words
bool isk_train_fwdtail_ok(const TrainShape& s);
int  isk_train_fwdtail_counters(long long maxB);
void isk_launch_train_fwdtail(const TrainShape& s, const TrainBuffers& b, const int* perm, int B, float inv_div, unsigned* cnt, hipStream_t st);
// cotangent mode: b.target holds the cotangent [B][nout]; delta_L = cot .* act'(z_L), no loss, no weights
struct OptArgs;
// fuse: the optimizer step rides on the kernel (see isk_train_grad1n_can_fuse); nullptr: the gradient goes to b.grad
void isk_launch_train_grad1n(const TrainShape& s, const TrainBuffers& b, const int* perm, int B, int nslab, hipStream_t st, const OptArgs* fuse = nullptr);
bool isk_train_grad1n_can_fuse(const TrainShape& s, int Fp, int B, int nslab, int nw8);
// ---------------------------------------------------------------- narrow tails (train_tailn.hpp) ----
#define TN_MAXL 3                   /* tail layers of the register-resident tail */
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
TailNDesc isk_tailn_desc(const TrainShape& s);
// ---------------------------------------------------------------- persistent training epoch (kernels_train_epoch.hip) ----
struct EpochPlan { int cps, KS, nt, G, U, zrows, nC, nS; size_t lds; };
#define ISK_EP_CNT_WORDS 4096         /* one 128-byte line per counter */
struct EpochArgs {
    const float* X; int Fp; const int* perm; int B, nb;         // batch rows of step s: perm[s*B .. s*B + B)
    float* theta; float* m; float* v;                            // engine (native) order
    long long off_lns, off_lnb, off_W0, off_b0;
    int ln;                                                      // leading LayerNorm
    int nt, nch, cps, KS, U, G, zrows, nC, nS;
    long long nparams;
    TailNDesc d;
    const float* target; const float* lossw; float inv_div;
    float* zp;                                                   // [KS][zrows][H1p]  K-slice partials of z1
    float* delta1;                                               // [zrows][H1p]      natural column order
    float* part;                                                 // [U][PS]           per-unit partial sums of the small gradients, [0] loss
    float* lnpart;                                               // [KS][nt][2][cps*16] LayerNorm-gradient shares of the tile owners
    unsigned* cnt;                                               // [ISK_EP_CNT_WORDS] hand-off counters (zeroed by the launcher)
    int opt_kind; float lr, reg, beta1, beta2, eps, rho;
    double bt1, bt2;                                             // beta^t of the epoch's first step
    int collapsed_in;
    double* loss_acc; double* host_pub;
    unsigned long long* dbg;                                     // diagnostics (option debug_stamps): [2][32] phase stamps of step 1, or nullptr
};
// false: the shape is not covered (the caller keeps the four-kernel chain).  ncu: compute units (the grid must be co-resident)
bool isk_train_epoch_plan(const TrainShape& s, int Fp, int B, int ncu, EpochPlan* p);
int  isk_launch_train_epoch(const TrainShape& s, const EpochPlan& p, EpochArgs a, hipStream_t st);
// kernels_train_chain.hip: the same for batches of at most 128 rows on the workgroups of ONE XCD (option train_persist = 2)
bool isk_train_chain_plan(const TrainShape& s, int Fp, int B, EpochPlan* p);
int  isk_launch_train_chain(const TrainShape& s, const EpochPlan& p, EpochArgs a, hipStream_t st);
#define ISK_MAX_SLABS 8
int  isk_train_tail_nblocks(int B);
long long isk_train_part_floats(const TrainShape& s, int maxB);

struct OptArgs {            // kept within two 64-byte kernarg lines
    float* theta; const float* grad; float* m; float* v;
    const double* bt;       // optional: device [beta1^t, beta2^t] of THIS step (then bc1/bc2 are derived from it, so the launch
                            // arguments do not change from step to step and the epoch can be replayed as a hipGraph)
    double* loss_acc;       // [0] epoch loss sum, [1] collapsed flag (set: skip the update, src/iso.jl:186-189 throws before update!),
                            // [2] (as int) the target's status word, [3] ranks that flagged an error in a step of this epoch
    double* host_pub;       // optional device pointer to MAPPED pinned host memory: [loss sum, collapsed flag, status word, peer flag]
                            // grad[n + 1] is the (all-reduced) rank-error flag: non-zero skips the update like a non-finite loss
    long long n;
    long long slab_prefix;  // gradient slabs are summed for parameter indices < slab_prefix
    int kind, nslab;
    float lr, reg, beta1, beta2, eps, rho;
    float bc1, bc2;         // 1 - beta1^t, 1 - beta2^t for this step (host copy, used when bt == nullptr)
    float gscale;           // gradient multiplier (1, or 1/global_batch in data-parallel mode)
};
void isk_launch_opt_step(const OptArgs& a, hipStream_t st);
// the same step with the gradient summed over the replicas of a single-process data-parallel group: g[r] is rank r's
// [gradient | loss | flag] buffer (peer-mapped device memory), summed in rank order -- the same bits on every replica
#define ISK_DP_MAXPEER 16
struct OptPeers { const float* g[ISK_DP_MAXPEER]; int n; };
void isk_launch_opt_step_peers(const OptArgs& a, const OptPeers& p, hipStream_t st);

// ---------------------------------------------------------------- coordinate pullback ----
// grad[n] = (d chi / d coords_n)' * cot[n]  (Zygote pullback of chicoords; the counterpart of the reference's
// sqpairdist_bwd_kernel!, src/utils/pairdists.jl:153-196, extended through LayerNorm and the MLP)
struct VjpArgs {
    const float* delta1;    // [Bpad][H1p] dL/dz1 from the tail kernel (cotangent mode)
    const float* xhat;      // [S][Fp]
    const float* raw;       // [S][Fp] distances (pdist featurizers) -- unused for identity
    const float* stats;     // [S][2]
    const float* coords;    // [S][D]
    const int2*  pairs;     // [F] or nullptr (identity featurizer: the gradient is w.r.t. the features)
    const float* theta;     // flat parameters
    long long off_W0, off_lns;   // offsets of W1 and the LayerNorm scale (off_lns < 0: no LayerNorm)
    float* grad;            // [S][D]
    long long S;
    int D, F, Fp, H1, H1p, ln_mode;
    int native;             // 1: theta in the engine's fused-path layout (fragment-linear W1), delta1 columns in isk_delta1_pos order
};
void isk_launch_vjp_back(const VjpArgs& a, hipStream_t st);

// The kernels of the target and of the training steps are short, dependent launches with small grids; the next
// iteration's featurizer runs beside them (side stream, one workgroup per CU) and its waves are busy with VALU / LDS work
// nearly every cycle.  Sharing a SIMD with one of them on equal terms doubled the length of a training step (60 -> 112 us
// on C3), so the latency-chain kernels raise their wave priority: the SIMD's issue arbiter serves them first and the
// featurizer fills the cycles they leave.
#define ISK_LATENCY_CHAIN_PRIO() __builtin_amdgcn_s_setprio(3)
