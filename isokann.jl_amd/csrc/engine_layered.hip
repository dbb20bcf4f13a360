// Layered (wide-network) path of the engine: networks whose layers do not fit the fused kernels (first hidden width > 160,
// or a tail too wide for LDS -- e.g. C5's densenet [2000,512,512,512,1]) run layer by layer through HBM.
#include <dlfcn.h>
#include <mutex>
#include "engine.hpp"

// bf16 planes per f32 operand of the layer-1 / dense-block forward GEMM: 0 = f32 MFMA, 3 = split-bf16, 1 = plain bf16
int bf16_planes(const isk_engine* e) { return e->cfg.forward_mode == 1 ? 3 : (e->cfg.forward_mode == 2 ? 1 : 0); }

// layer-by-layer forward: every layer is a sequence of dense-block launches of the same GEMM kernels; hidden
// activations [S][pad32(H)] ping-pong between actA / actB (zero-initialised: the padding columns stay 0)
static void wide_args_shape(const isk_engine* e, WideFwdArgs& a) {
    a.Fp = e->Fp; a.nhidden = e->L - 1;
    for (int l = 0; l + 1 < e->L; ++l) { a.N[l] = e->cfg.widths[l + 1]; a.Kp[l] = isk_pad32(e->cfg.widths[l]); }
    a.Klast = e->cfg.widths[e->L - 1]; a.nout = e->nout; a.x_bf16 = 1; a.ldx = isk_ld16(e->Fp);
}
bool wide_forward_shape_ok(const isk_engine* e) {
    WideFwdArgs a = {};
    wide_args_shape(e, a);
    if (e->L < 2 || !isk_wide_fwd_ok(a)) return false;
    for (int l = 0; l + 1 < e->L; ++l) if (!e->b16.Wf[l]) return false;
    return true;
}

// plain-bf16 forward (forward_mode 2) of a network whose hidden layers are <= 512 wide: ONE fused kernel, activations in LDS
static int forward_wide_bf16(isk_engine* e, const float* X, long long S, float* out) {
    WideFwdArgs a = {};
    a.X = X; a.x_bf16 = e->xh16; a.S = S; a.Fp = e->Fp; a.ldx = isk_ld16(e->Fp); a.nhidden = e->L - 1;
    for (int l = 0; l + 1 < e->L; ++l) {
        a.N[l] = e->cfg.widths[l + 1]; a.Kp[l] = isk_pad32(e->cfg.widths[l]); a.Wt[l] = e->b16.Wf[l];
        a.bias[l] = (l == 0 && e->cfg.layernorm) ? e->d_bias[0] : e->d_theta + e->off_b[l];
    }
    a.act = e->cfg.activation;
    a.Wlast = e->d_theta + e->off_W[e->L - 1]; a.blast = e->d_theta + e->off_b[e->L - 1];
    a.Klast = e->cfg.widths[e->L - 1]; a.nout = e->nout; a.lastact = e->cfg.lastactivation;
    a.out = out; a.out_ld = e->nout;
    if (!wide_forward_shape_ok(e) || !isk_wide_fwd_ok(a)) return ISK_EUNSUPPORTED;
    bf16_weight_images_ensure(e, false, true);
    return isk_launch_wide_fwd(a, e->stream);
}

int forward_layered(isk_engine* e, const float* X, long long S, float* out, float* actA, float* actB) {
    // (feature rows written as bf16 can only be read by the fused kernel: xh16 was decided with them)
    if (e->xh16) return forward_wide_bf16(e, X, S, out);
    const float* in = X;
    int in_ld = e->Fp;
    for (int l = 0; l < e->L; ++l) {
        const bool last = (l == e->L - 1);
        const int N = e->cfg.widths[l + 1];
        const int nt_total = isk_pad16(N) / 16;
        float* dst = last ? out : ((l % 2 == 0) ? actA : actB);
        const int out_ld = last ? e->nout : isk_pad32(N);
        const int planes = bf16_planes(e);
        if (planes && e->w16_dirty)
            isk_launch_fold_bf16(e->d_Wt[l], e->d_W16L[l], layer_kp(e, l), nt_total, planes, e->stream);
        // plain bf16 keeps 16 output tiles per workgroup (two launches for a 512-wide layer), the others 8
        const int tb = planes == 1 ? ISK_MAX_NT_BF16 : 8;
        // equal column blocks of a bf16 layer go out as ONE grouped launch (the sample rows are then read from HBM once
        // per layer instead of once per column block); ISK_DENSE_GROUPS=0 restores one launch per block
        const int grouped = e->opt.dense_groups;
        const int groups = (grouped && planes && nt_total > tb && nt_total % tb == 0 && nt_total / tb <= 8) ? nt_total / tb : 1;
        auto launch_block = [&](int t0, int ntb, int ngroups) -> int {
            MlpFwdArgs a;
            std::memset(&a, 0, sizeof(a));
            a.X = in; a.S = S; a.Fp = in_ld; a.nchunks = in_ld / 16;
            a.W1t = e->d_Wt[l]; a.bias1 = e->d_bias[l] + t0 * 16;
            a.act1 = act_of_layer(e, l); a.H1 = N; a.nt1 = ntb; a.ntail = 0;
            a.out = dst; a.nout = out_ld;
            a.dense_mode = 1; a.w_nt_total = nt_total; a.w_tile0 = t0; a.out_ld = out_ld; a.out_col0 = t0 * 16;
            a.dense_groups = ngroups;
            a.stride_a = ntb * 16 + 4; a.stride_b = 4;
            isk_fwd_layout_biases(a);
            int r = ISK_EUNSUPPORTED;
            if (planes) r = isk_launch_mlp_fwd_bf16x3(a, e->d_W16L[l], planes, e->stream);
            if (r != 0 && ngroups == 1 && ntb <= ISK_MAX_NT) r = isk_launch_mlp_fwd(a, e->stream);     // the f32 kernel: one block per launch
            return r;
        };
        for (int t0 = 0; t0 < nt_total; t0 += tb * groups) {
            const int ntb = std::min(tb, nt_total - t0);
            int r = launch_block(t0, ntb, groups);
            if (r != 0 && groups > 1)
                for (int g = 0; g < groups; ++g)
                    if ((r = launch_block(t0 + g * tb, tb, 1)) != 0) break;
            if (r != 0) return r;
        }
        in = dst;
        in_ld = out_ld;
    }
    e->w16_dirty = false;
    return 0;
}

// ---------------------------------------------------------------------------------------
// bf16 MFMA training GEMMs (option train_bf16): the arithmetic BASELINE.json names for the wide C5 network -- bf16 operands,
// f32 accumulate, f32 master weights and optimizer.  kernels_gemm16.hip explains the operand images.
// ---------------------------------------------------------------------------------------
bool bf16_train_active(const isk_engine* e) { return e->layered && e->opt.train_bf16 != 0; }
// ... and the same kernels in Float32 arithmetic (f32 fragments, v_mfma_f32_16x16x4_f32, f32 images): the DEFAULT of the layered
// training path since round 4 (round 5: the vendor-library sgemm route it had overtaken is gone)
bool mfma32_train_active(const isk_engine* e) { return e->layered && !e->opt.train_bf16; }

// the per-batch images (the weight images belong to the engine, not to the data)
void bf16_train_release(isk_engine* e) {
    auto& b = e->b16;
    free_dev(b.xb); free_dev(b.xt); free_dev(b.G0); free_dev(b.head_part);
    b.xb = b.xt = nullptr; b.G0 = nullptr; b.head_part = nullptr; b.head_stride = 0;
    for (int l = 0; l < ISK_MAX_LAYERS; ++l) {
        free_dev(b.a16[l]); free_dev(b.at16[l]); free_dev(b.d16[l]); free_dev(b.dt16[l]);
        b.a16[l] = b.at16[l] = b.d16[l] = b.dt16[l] = nullptr;
    }
    b.rows = 0; b.esz = 0;
    free_dev(b.xb_all); free_dev(b.xt_all);
    b.xb_all = b.xt_all = nullptr; b.xb_all_bytes = b.xt_all_bytes = 0; b.ep_nb = 0;
}

int bf16_weight_images_alloc(isk_engine* e) {
    for (int l = 0; l < e->L; ++l) {
        const size_t bytes = (size_t)std::max(e->cfg.widths[l + 1], 512) * isk_pad32(e->cfg.widths[l]) * 4;    // rows >= N stay zero; sized for f32 elements
        if (hipMalloc(&e->b16.Wt[l], bytes) != hipSuccess) return ISK_ENOMEM;
        (void)hipMemset(e->b16.Wt[l], 0, bytes);
        if (l + 1 < e->L && e->cfg.widths[l + 1] <= 512) {           // hidden layer the fused wide forward can take: 512 (padded) columns
            const size_t fb = (size_t)512 * ((e->cfg.widths[l] + 127) / 128 * 128) * 2;     // zero steps up to a multiple of four
            if (hipMalloc(&e->b16.Wf[l], fb) != hipSuccess) return ISK_ENOMEM;
            (void)hipMemset(e->b16.Wf[l], 0, fb);
        }
    }
    e->b16.wt_dirty = e->b16.wf_dirty = true;
    return ISK_OK;
}

// the row-major images feed the training GEMMs (rebuilt every optimizer step), the fragment-linear ones the fused forward
// (rebuilt once per forward after a parameter change)
static bool weight_image_args(isk_engine* e, bool want_wt, bool want_wf, WtImagesArgs& w, int wt_esz = 2) {
    auto& b = e->b16;
    if (want_wt && b.wt_esz != wt_esz) b.wt_dirty = true;            // the images hold the other element type
    want_wt = want_wt && b.wt_dirty && b.Wt[0];
    want_wf = want_wf && b.wf_dirty && b.Wf[0];
    if (!want_wt && !want_wf) return false;
    w = WtImagesArgs{};
    w.theta = e->d_theta; w.scale0 = e->cfg.layernorm ? e->d_theta + e->off_lns : nullptr; w.nlayers = e->L;
    for (int l = 0; l < e->L; ++l) {
        w.off_W[l] = e->off_W[l]; w.K[l] = e->cfg.widths[l]; w.N[l] = e->cfg.widths[l + 1]; w.Kp[l] = isk_pad32(w.K[l]);
        w.Wt[l] = want_wt ? b.Wt[l] : nullptr;
        w.Wf[l] = want_wf ? b.Wf[l] : nullptr;
    }
    w.wt_f32 = wt_esz == 4;
    if (want_wt) { b.wt_dirty = false; b.wt_esz = wt_esz; }
    if (want_wf) b.wf_dirty = false;
    return true;
}
void bf16_weight_images_ensure(isk_engine* e, bool want_wt, bool want_wf) {
    WtImagesArgs w;
    if (weight_image_args(e, want_wt, want_wf, w)) isk_launch_wt_images(w, e->stream);
}

__global__ void fill_u16_kernel(unsigned short* p, long long n, unsigned short v) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = v;
}
__global__ void fill_f32_kernel(float* p, long long n, float v) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = v;
}

// esz: bytes per image element (2 = bf16, 4 = Float32 arithmetic)
static bool bf16_train_ensure(isk_engine* e, long long Bp, int esz) {
    auto& b = e->b16;
    if (b.rows >= Bp && b.esz == esz) return true;
    (void)hipStreamSynchronize(e->stream);
    bf16_train_release(e);
    const long long rows = isk_pad32((int)std::max<long long>(Bp, std::min<long long>(e->N, 4096)));     // usual batches without re-allocation
    const int L = e->L;
    bool ok = true;
    auto alloc = [&](void** p, size_t bytes) { if (ok && hipMalloc(p, bytes) != hipSuccess) ok = false; if (ok) (void)hipMemsetAsync(*p, 0, bytes, e->stream); };
    alloc(&b.xb, (size_t)rows * e->Fp * esz);
    alloc(&b.xt, (size_t)(e->F + 1) * rows * esz);
    for (int l = 0; l < L && ok; ++l) {
        const int K = e->cfg.widths[l], N = e->cfg.widths[l + 1];
        alloc(&b.a16[l], (size_t)rows * isk_pad32(N) * esz);
        alloc(&b.at16[l], (size_t)(N + 1) * rows * esz);
        alloc(&b.d16[l], (size_t)rows * isk_pad32(N) * esz);
        alloc(&b.dt16[l], (size_t)N * rows * esz);
        (void)K;
        // the row of ones behind the N activation rows
        if (ok && esz == 2) hipLaunchKernelGGL(fill_u16_kernel, dim3(64), dim3(256), 0, e->stream, (unsigned short*)b.at16[l] + (size_t)N * rows, rows,
                                               (unsigned short)0x3F80);
        if (ok && esz == 4) hipLaunchKernelGGL(fill_f32_kernel, dim3(64), dim3(256), 0, e->stream, (float*)b.at16[l] + (size_t)N * rows, rows, 1.0f);
    }
    if (ok && e->cfg.layernorm) alloc((void**)&b.G0, (size_t)(e->F + 1) * e->cfg.widths[1] * sizeof(float));
    if (ok && L >= 2) {
        b.head_stride = ((long long)e->cfg.widths[L - 1] * e->nout + e->nout + 1 + 3) & ~3ll;
        alloc((void**)&b.head_part, (size_t)ISK_HEAD_MAXG * b.head_stride * sizeof(float));
    }
    if (!ok) { bf16_train_release(e); return false; }
    b.rows = rows; b.esz = esz;
    return true;
}

// The batch images (row gather + transposed copy with the row of ones) of EVERY batch of the coming epoch, in one launch: they
// depend on the shuffle order only, and in front of each step's chain the gather was 9..19 us of its ~150.  Valid until
// layered_epoch_end; a step whose rows are not a batch of this epoch (isk_batch_grads, another batch size) gathers for itself.
void layered_epoch_gather(isk_engine* e, const int* d_perm, long long bs, long long nb) {
    auto& b = e->b16;
    b.ep_nb = 0;
#ifndef ISK_EXPERIMENTAL
    (void)d_perm; (void)bs; (void)nb;            // (measured: no faster than the per-step gather; built with -DISK_EXPERIMENTAL only)
#else
    if (!e->layered || !e->opt.epoch_gather || nb < 2 || !(bf16_train_active(e) || mfma32_train_active(e))) return;
    const int esz = bf16_train_active(e) ? 2 : 4;
    const long long Bp = isk_pad32((int)bs);
    const size_t need_b = (size_t)nb * Bp * e->Fp * esz, need_t = (size_t)nb * (e->F + 1) * Bp * esz;
    if (need_b + need_t > ((size_t)2 << 30)) return;                     // (a very long epoch of a very wide input: gather per step)
    if (need_b > b.xb_all_bytes || need_t > b.xt_all_bytes) {
        (void)hipStreamSynchronize(e->stream);
        if (b.xb_all) (void)hipFree(b.xb_all);
        if (b.xt_all) (void)hipFree(b.xt_all);
        b.xb_all = b.xt_all = nullptr; b.xb_all_bytes = b.xt_all_bytes = 0;
        if (hipMalloc(&b.xb_all, need_b) != hipSuccess || hipMalloc(&b.xt_all, need_t) != hipSuccess) {
            (void)hipGetLastError();
            if (b.xb_all) (void)hipFree(b.xb_all);
            b.xb_all = b.xt_all = nullptr;
            return;
        }
        b.xb_all_bytes = need_b; b.xt_all_bytes = need_t;
    }
    isk_launch_gather_epoch(e->d_xs32, e->Fp, d_perm, (int)bs, (int)Bp, e->F, e->Fp, b.xb_all, b.xt_all, Bp, esz == 4, (int)nb,
                            (long long)Bp * e->Fp * esz, (long long)(e->F + 1) * Bp * esz, e->stream);
    b.ep_perm = d_perm; b.ep_bs = bs; b.ep_nb = nb; b.ep_esz = esz;
#endif
}
void layered_epoch_end(isk_engine* e) { e->b16.ep_nb = 0; }

// The boundary between two optimizer steps as ONE launch (option fuse_boundary; kernels_gemm16.hip opt_prep_kernel): the optimizer
// step, the weight images of the next step, and -- when the epoch loop has announced them (e->next_idx) -- the batch images of the
// next step.  false: not applicable here (the caller launches opt_step; the next step prepares its own images).
bool layered_opt_boundary(isk_engine* e, const OptArgs& a) {
    auto& b = e->b16;
    b.pre_idx = nullptr;
    if (!e->layered || !e->opt.fuse_boundary || e->cfg.layernorm || !(bf16_train_active(e) || mfma32_train_active(e))) return false;
    if (b.rows <= 0 || !b.Wt[0] || !b.xb || a.nslab != 1) return false;
    const int esz = bf16_train_active(e) ? 2 : 4;
    if (b.esz != esz) return false;
    WtImagesArgs w;
    b.wt_dirty = true;                                               // (the step changes every weight)
    if (!weight_image_args(e, true, false, w, esz)) return false;
    const int* next = e->next_idx;
    const int B = e->next_B, Bp = isk_pad32(B);
    if (next && (B < 1 || Bp > b.rows)) next = nullptr;
    isk_launch_opt_prep(a, w, e->off_b, e->d_xs32, e->Fp, next, B, Bp, e->F, e->Fp, b.xb, b.xt, b.rows, esz == 4, e->stream);
    if (next) { b.pre_idx = next; b.pre_B = B; b.pre_esz = esz; }
    return true;
}

// f32math: the same chain of fused products in Float32 arithmetic (f32 images, v_mfma_f32_16x16x4_f32)
static bool launch_grads_layered_bf16(isk_engine* e, const int* d_idx, int B, float inv_div, const int f32math) {
    const int L = e->L;
    const int Bp = isk_pad32(B);
    const int esz = f32math ? 4 : 2;
    if (!bf16_train_ensure(e, Bp, esz)) return false;
    auto& b = e->b16;
    const long long ldB = b.rows;                    // row length of the transposed images
    hipStream_t st = e->stream;
    const bool ln = e->cfg.layernorm != 0;
    const void* xb = b.xb;                          // this batch's images: [Bp][Fp] and [F + 1][ldX]
    const void* xt = b.xt;
    long long ldX = ldB;
    {
        // the batch images and (behind an optimizer step: always) the row-major weight images, one launch -- unless the batch
        // is one of an epoch whose images were gathered up front (layered_epoch_gather)
        WtImagesArgs w;
        const bool dirty = weight_image_args(e, true, false, w, esz);
        const bool pre = b.pre_idx && b.pre_idx == d_idx && b.pre_B == B && b.pre_esz == esz;    // written by the last step-boundary launch
        b.pre_idx = nullptr;
        const long long off = pre ? -1 : b.ep_nb > 0 && B == b.ep_bs && esz == b.ep_esz && d_idx >= b.ep_perm ? (long long)(d_idx - b.ep_perm) : -1;
        if (off >= 0 && off % B == 0 && off / B < b.ep_nb) {
            const long long k = off / B;
            xb = (const char*)b.xb_all + (size_t)k * Bp * e->Fp * esz;
            xt = (const char*)b.xt_all + (size_t)k * (e->F + 1) * Bp * esz;
            ldX = Bp;
            if (dirty) isk_launch_wt_images(w, st);
        } else if (pre) {
            if (dirty) isk_launch_wt_images(w, st);                  // (parameters set from outside since: the batch images stand)
        } else if (dirty) {
            isk_launch_prep_images(e->d_xs32, e->Fp, d_idx, B, Bp, e->F, e->Fp, b.xb, b.xt, ldB, f32math, w, st);
        } else {
            isk_launch_gather_images(e->d_xs32, e->Fp, d_idx, B, Bp, e->F, e->Fp, b.xb, b.xt, ldB, f32math, st);
        }
    }
    bool ok = true;
    // the head of the backward pass as one launch (kernels_gemm16.hip head_kernel) where the network has >= 3 layers and <= 4 outputs
    const int Kh = e->cfg.widths[L - 1];
    const bool head = e->opt.fuse_head && L >= 3 && b.head_part && isk_head_ok(Kh, e->nout, B, Bp, isk_pad32(Kh), isk_pad16(Kh));
    for (int l = 0; l < (head ? L - 1 : L) && ok; ++l) {
        const int K = e->cfg.widths[l], N = e->cfg.widths[l + 1];
        Gemm16Args g = {};
        g.A = l == 0 ? xb : b.a16[l - 1]; g.lda = l == 0 ? e->Fp : isk_pad32(K); g.a_f32 = f32math;
        g.B = b.Wt[l]; g.ldb = isk_pad32(K); g.b_f32 = f32math; g.f32math = f32math;
        g.M = B; g.Nn = N; g.Kc = isk_pad32(K); g.kvalid = g.Kc; g.waves = e->opt.g16_waves;
        g.epi = ISK_G16_FWD;
        g.bias = (l == 0 && ln) ? e->d_bias[0] : e->d_theta + e->off_b[l];
        g.act = l < L - 1 ? act_of_layer(e, l) : ISK_ACT_IDENTITY;          // the last layer's activation is applied by the loss kernel
        g.z = e->d_lz[l]; g.ldz = isk_pad16(N);
        if (l < L - 1) { g.o16 = b.a16[l]; g.ld16 = isk_pad32(N); g.M16 = Bp; g.ot16 = b.at16[l]; g.ldt = (int)ldB; }
        ok = isk_launch_gemm16(g, st) == 0;
    }
    if (!ok) return false;
    // (the last layer + loss + both deltas as ONE "head" kernel was built and measured: 59 us against the 45 us of the four launches
    // it replaced -- eight rows per wave are eight dependent round trips; not kept)
    Gemm16Head hj = {};
    if (head) {
        isk_launch_head(b.a16[L - 2], isk_pad32(Kh), f32math, e->d_lz[L - 2], isk_pad16(Kh), act_of_layer(e, L - 2), e->d_theta + e->off_W[L - 1],
                        e->d_theta + e->off_b[L - 1], Kh, e->nout, e->d_target, e->d_lossw, d_idx, B, Bp, inv_div, e->cfg.lastactivation,
                        b.d16[L - 2], isk_pad32(Kh), b.dt16[L - 2], ldB, f32math, b.head_part, b.head_stride, st);
        hj.part = b.head_part; hj.stride = b.head_stride; hj.G = Bp / ISK_HEAD_ROWS; hj.len = (long long)Kh * e->nout + e->nout;
        hj.out = e->d_grad + e->off_W[L - 1];                            // W_L then b_L: Flux keeps the bias right behind the weights
        hj.loss = e->d_grad + e->nparams; hj.rank_err = e->d_rank_err;
    } else {
        isk_launch_loss_delta16(e->d_lz[L - 1], isk_pad16(e->nout), e->d_target, e->d_lossw, d_idx, B, Bp, e->nout, inv_div, e->cfg.lastactivation,
                                b.d16[L - 1], isk_pad32(e->nout), b.dt16[L - 1], ldB, f32math, e->d_grad + e->nparams, e->d_rank_err, st);
    }
    for (int l = head ? L - 2 : L - 1; l >= 0 && ok; --l) {
        const int K = e->cfg.widths[l], N = e->cfg.widths[l + 1];
        const bool ln0 = l == 0 && ln;
        Gemm16Args pair[3] = {};
        Gemm16Args& g = pair[0];
        g.A = l == 0 ? xt : b.at16[l - 1]; g.lda = l == 0 ? ldX : ldB; g.a_f32 = f32math;
        g.B = b.dt16[l]; g.ldb = ldB; g.b_f32 = f32math; g.f32math = f32math;
        g.M = K + 1; g.Nn = N; g.Kc = Bp; g.kvalid = Bp; g.waves = e->opt.g16_waves;
        g.epi = ISK_G16_DW;
        g.gout = ln0 ? b.G0 : e->d_grad + e->off_W[l]; g.gld = N;           // row K lands on b_l: Flux keeps it right behind W_l
        int np = 1;
        if (l > 0) {
            np = 2;
            Gemm16Args& x = pair[1];
            x.A = b.d16[l]; x.lda = isk_pad32(N); x.a_f32 = f32math; x.f32math = f32math;
            x.B = e->d_theta + e->off_W[l]; x.ldb = N; x.b_f32 = 1;           // Flux (out, in) column-major == [k][n]: n contiguous
            x.M = B; x.Nn = K; x.Kc = isk_pad32(N); x.kvalid = N; x.waves = e->opt.g16_waves;
            x.epi = ISK_G16_DX;
            x.zprev = e->d_lz[l - 1]; x.ldzp = isk_pad16(K); x.act = act_of_layer(e, l - 1);
            x.o16 = b.d16[l - 1]; x.ld16 = isk_pad32(K); x.M16 = Bp; x.ot16 = b.dt16[l - 1]; x.ldt = (int)ldB;
        }
        // dW_l + db_l and delta_{l-1} both need only delta_l: one launch
        ok = isk_launch_gemm16_group(pair, np, st, head && l == L - 2 ? &hj : nullptr) == 0;
        // every gradient block of the layers >= 1 is now enqueued (Flux order: they are one contiguous range behind layer 0's):
        // the data-parallel driver sends them on their way while the first layer's products run
        if (ok && l == 1 && e->grad_hook) e->grad_hook(e);
        if (ln0) {
            (void)hipMemcpyAsync(e->d_grad + e->off_b[0], b.G0 + (size_t)K * N, (size_t)N * sizeof(float), hipMemcpyDeviceToDevice, st);
            isk_launch_ln_unfold(b.G0, b.G0 + (size_t)K * N, e->d_theta, e->off_W[0], e->off_lns, e->off_lnb, e->F, N, e->d_grad, st);
        }
    }
    e->last_nslab = 1;
    return ok;
}

// layered path: one optimizer step's gradient, layer by layer (see kernels_layered.hip)
void launch_grads_layered(isk_engine* e, const int* d_idx, int B, float inv_div) {
    if (bf16_train_active(e) && launch_grads_layered_bf16(e, d_idx, B, inv_div, 0)) return;
    if (mfma32_train_active(e) && launch_grads_layered_bf16(e, d_idx, B, inv_div, 1)) return;
    // fallback when the GEMM images cannot be allocated: the row-GEMM / tile-dW kernels of kernels_layered.hip
    ensure_folded(e);                // they read the f32 weight images (an epoch on the MFMA GEMMs leaves them unfolded)
    const int L = e->L;
    const long long Bpad = ((B + 15) / 16) * 16;
    const float* xs = e->d_xs32;
    hipStream_t st = e->stream;
    for (int l = 0; l < L; ++l) {
        const int N = e->cfg.widths[l + 1], Np = isk_pad16(N);
        const float* X = (l == 0) ? xs : e->d_la[l - 1];
        const int ldx = (l == 0) ? e->Fp : isk_pad16(e->cfg.widths[l]);
        isk_launch_gemm_rows(X, ldx, e->d_Wt[l], e->d_bias[l], Np / 16, l == 0 ? d_idx : nullptr, B, e->d_lz[l], st);
        if (l < L - 1) isk_launch_act(e->d_lz[l], e->d_la[l], Bpad, Np, N, act_of_layer(e, l), st);
    }
    const int nop = isk_pad16(e->nout);
    isk_launch_loss_delta(e->d_lz[L - 1], nop, e->d_target, e->d_lossw, d_idx, B, e->nout, inv_div, e->cfg.lastactivation,
                          e->d_ld[L - 1], nop, e->d_grad + e->nparams, e->d_rank_err, st);
    for (int l = L - 1; l >= 0; --l) {
        const int K = e->cfg.widths[l], N = e->cfg.widths[l + 1], Np = isk_pad16(N);
        const float* X = (l == 0) ? xs : e->d_la[l - 1];
        const int ldx = (l == 0) ? e->Fp : isk_pad16(K);
        const bool ln0 = (l == 0 && e->cfg.layernorm);
        isk_launch_dense_dw(X, ldx, l == 0 ? d_idx : nullptr, e->d_ld[l], Np, B, K, N, ln0 ? e->d_G0 : e->d_grad + e->off_W[l],
                            e->d_grad + e->off_b[l], st);
        if (ln0)
            isk_launch_ln_unfold(e->d_G0, e->d_grad + e->off_b[0], e->d_theta, e->off_W[0], e->off_lns, e->off_lnb, e->F, N, e->d_grad, st);
        if (l > 0) {
            // delta_{l-1} = (delta_l W_l) .* act'(z_{l-1});  a_{l-1} is dead after dW_l and receives the raw product
            isk_launch_gemm_rows(e->d_ld[l], Np, e->d_WtT[l], e->d_zero, isk_pad16(K) / 16, nullptr, B, e->d_la[l - 1], st);
            isk_launch_dact_mul(e->d_la[l - 1], e->d_lz[l - 1], e->d_ld[l - 1], Bpad, isk_pad16(K), K, act_of_layer(e, l - 1), st);
        }
    }
    e->last_nslab = 1;
}
