// Evaluation entry points on caller-provided columns (not the resident data): the featurizer, the model call, chicoords,
// the Koopman expectation and validation loss of other coordinates, and the coordinate pullback.
#include "engine.hpp"

extern "C" int isk_features(isk_engine* e, const float* coords, int64_t D, int64_t ncols, float* out) {
    if (!e || !coords || !out) return fail(ISK_EINVAL, "null argument");
    if (D != e->Din) return fail(ISK_EINVAL, "coordinate dimension mismatch");
    if (ncols <= 0) return ISK_OK;
    HIPCHK(hipSetDevice(e->cfg.device));
    TmpBuf in, xh;
    HIPCHK(hipMalloc(&in.p, (size_t)ncols * D * sizeof(float)));
    HIPCHK(hipMalloc(&xh.p, (size_t)ncols * e->Fp * sizeof(float)));
    HIPCHK(hipMemcpyAsync(in.p, coords, (size_t)ncols * D * sizeof(float), hipMemcpyHostToDevice, e->stream));
    FeatArgs a = {};
    a.in = (const float*)in.p; a.pairs = e->d_pairs; a.packed_pairs = e->d_packed; a.xhat = (float*)xh.p; a.S = ncols; a.D = e->Din; a.F = e->F; a.Fp = e->Fp;
    a.normalize = 0; a.ln_mode = e->cfg.ln_mode; a.eps = e->cfg.ln_eps;   // raw features: the featurizer alone
    a.background = 0;
    isk_launch_featurize(a, e->stream);
    HIPCHK(hipMemcpy2DAsync(out, (size_t)e->F * sizeof(float), xh.p, (size_t)e->Fp * sizeof(float), (size_t)e->F * sizeof(float),
                            (size_t)ncols, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipGetLastError());
    return ISK_OK;
}

// x rows already on the device as raw features [ncols][F] or coords [ncols][Din]
static int forward_cols(isk_engine* e, const float* host_in, int64_t rowlen, int64_t ncols, bool is_coords, float* out) {
    TmpBuf in, xh, o, xsc;
    HIPCHK(hipMalloc(&in.p, (size_t)ncols * rowlen * sizeof(float)));
    HIPCHK(hipMalloc(&xh.p, (size_t)isk_plane_rows(ncols) * e->Fp * sizeof(float)));
    HIPCHK(hipMalloc(&o.p, (size_t)ncols * e->nout * sizeof(float)));
    if (e->planes) HIPCHK(hipMalloc(&xsc.p, (size_t)ncols * sizeof(float)));
    HIPCHK(hipMemcpyAsync(in.p, host_in, (size_t)ncols * rowlen * sizeof(float), hipMemcpyHostToDevice, e->stream));
    FeatArgs a = {};
    a.in = (const float*)in.p; a.pairs = is_coords ? e->d_pairs : nullptr; a.packed_pairs = is_coords ? e->d_packed : nullptr;
    a.xhat = (float*)xh.p; a.S = ncols;
    a.D = (int)rowlen; a.F = e->F; a.Fp = e->Fp;
    a.normalize = e->cfg.layernorm; a.ln_mode = e->cfg.ln_mode; a.eps = e->cfg.ln_eps;
    a.background = 0;
    a.planes = e->planes; a.xscale = (float*)xsc.p; a.out_bf16 = e->xh16; a.ld16 = isk_ld16(e->Fp);
    isk_launch_featurize(a, e->stream);
    MlpFwdArgs f = fwd_args(e, (const float*)xh.p, ncols, (float*)o.p);
    f.xscale = (const float*)xsc.p;
    if (launch_forward(e, f) != 0) return fail(ISK_EUNSUPPORTED, "model shape not covered by the forward kernel");
    HIPCHK(hipMemcpyAsync(out, o.p, (size_t)ncols * e->nout * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipGetLastError());
    return ISK_OK;
}

extern "C" int isk_forward(isk_engine* e, const float* x, int64_t F, int64_t ncols, float* out) {
    if (!e || !x || !out) return fail(ISK_EINVAL, "null argument");
    if (F != e->F) return fail(ISK_EINVAL, "feature dimension mismatch");
    if (ncols <= 0) return ISK_OK;
    HIPCHK(hipSetDevice(e->cfg.device));
    return forward_cols(e, x, F, ncols, false, out);
}

extern "C" int isk_chicoords(isk_engine* e, const float* coords, int64_t D, int64_t ncols, float* out) {
    if (!e || !coords || !out) return fail(ISK_EINVAL, "null argument");
    if (D != e->Din) return fail(ISK_EINVAL, "coordinate dimension mismatch");
    if (ncols <= 0) return ISK_OK;
    HIPCHK(hipSetDevice(e->cfg.device));
    return forward_cols(e, coords, D, ncols, e->cfg.featurizer != ISK_FEAT_IDENTITY, out);
}

// featurize + forward + K-mean on caller-provided coordinates; result stays in `ks_dev` ([ncols][nout]) when given
static int koopman_cols(isk_engine* e, const float* ys, int64_t D, int64_t K, int64_t ncols, float* out_host) {
    const bool is_coords = e->cfg.featurizer != ISK_FEAT_IDENTITY;
    const long long S = K * ncols;
    TmpBuf in, xh, o, ks, xsc;
    HIPCHK(hipMalloc(&in.p, (size_t)S * D * sizeof(float)));
    HIPCHK(hipMalloc(&xh.p, (size_t)isk_plane_rows(S) * e->Fp * sizeof(float)));
    if (e->planes) HIPCHK(hipMalloc(&xsc.p, (size_t)S * sizeof(float)));
    HIPCHK(hipMalloc(&o.p, (size_t)S * e->nout * sizeof(float)));
    HIPCHK(hipMalloc(&ks.p, (size_t)ncols * e->nout * sizeof(float)));
    HIPCHK(hipMemcpyAsync(in.p, ys, (size_t)S * D * sizeof(float), hipMemcpyHostToDevice, e->stream));
    FeatArgs a = {};
    a.in = (const float*)in.p; a.pairs = is_coords ? e->d_pairs : nullptr; a.packed_pairs = is_coords ? e->d_packed : nullptr;
    a.xhat = (float*)xh.p; a.S = S; a.D = (int)D; a.F = e->F; a.Fp = e->Fp;
    a.normalize = e->cfg.layernorm; a.ln_mode = e->cfg.ln_mode; a.eps = e->cfg.ln_eps; a.background = 0;
    a.planes = e->planes; a.xscale = (float*)xsc.p; a.out_bf16 = e->xh16; a.ld16 = isk_ld16(e->Fp);
    isk_launch_featurize(a, e->stream);
    MlpFwdArgs f = fwd_args(e, (const float*)xh.p, S, (float*)o.p);
    f.xscale = (const float*)xsc.p;
    if (launch_forward(e, f) != 0) return fail(ISK_EUNSUPPORTED, "model shape not covered by the forward kernel");
    KmeanArgs k;
    k.out_ys = (const float*)o.p; k.ks = (float*)ks.p; k.w = nullptr; k.N = ncols; k.K = (int)K; k.nout = e->nout; k.zero2 = nullptr;
    isk_launch_kmean(k, e->stream);
    HIPCHK(hipMemcpyAsync(out_host, ks.p, (size_t)ncols * e->nout * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipGetLastError());
    return ISK_OK;
}

extern "C" int isk_koopman_coords(isk_engine* e, const float* ys, int64_t D, int64_t K, int64_t ncols, float* out) {
    if (!e || !ys || !out) return fail(ISK_EINVAL, "null argument");
    if (D != e->Din) return fail(ISK_EINVAL, "coordinate dimension mismatch");
    if (K < 1) return fail(ISK_EINVAL, "K must be positive");
    if (ncols <= 0) return ISK_OK;
    HIPCHK(hipSetDevice(e->cfg.device));
    return koopman_cols(e, ys, D, K, ncols, out);
}

extern "C" int isk_validationloss(isk_engine* e, const float* vxs, const float* vys, int64_t D, int64_t K, int64_t Nv, double* loss) {
    CHK(need_data(e));
    if (!vxs || !vys || !loss || Nv < 1) return fail(ISK_EINVAL, "bad argument");
    if (e->nout != 1) return fail(ISK_EINVAL, "TransformShiftscale only works with one dimensional chi functions");
    if (D != e->Din) return fail(ISK_EINVAL, "coordinate dimension mismatch");
    std::vector<float> c(Nv), k1(Nv), k2(e->N);
    CHK(forward_cols(e, vxs, D, Nv, e->cfg.featurizer != ISK_FEAT_IDENTITY, c.data()));       // chi(vx)
    CHK(koopman_cols(e, vys, D, K, Nv, k1.data()));                                            // K chi on the validation pairs
    CHK(forward_resident(e, 1));                                                               // K chi on ALL training data
    launch_kmean(e);
    HIPCHK(hipMemcpyAsync(k2.data(), e->d_ks, (size_t)e->N * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    // Julia's extrema propagates NaN (then max > min is false and shiftscale throws): std::fmin / fmax would drop it
    float lo = INFINITY, hi = -INFINITY;
    bool nan = false;
    for (float v : k1) { nan |= std::isnan(v); lo = std::fmin(lo, v); hi = std::fmax(hi, v); }
    for (float v : k2) { nan |= std::isnan(v); lo = std::fmin(lo, v); hi = std::fmax(hi, v); }
    if (nan || !(hi > lo)) return fail(ISK_ECONSTANT_CHI, isk_status_string(ISK_ECONSTANT_CHI));
    double acc = 0;
    for (int64_t i = 0; i < Nv; ++i) {
        const float sk = (k1[i] - lo) / (hi - lo);
        const double d = (double)c[i] - (double)sk;
        acc += d * d;
    }
    *loss = acc / (double)Nv;
    return ISK_OK;
}

// chicoords pullback: grad (D, ncols) = (d chi / d coords)' * cot, column by column
extern "C" int isk_chicoords_pullback(isk_engine* e, const float* coords, int64_t D, int64_t ncols, const float* cot, float* grad) {
    if (!e || !coords || !cot || !grad) return fail(ISK_EINVAL, "null argument");
    if (D != e->Din) return fail(ISK_EINVAL, "coordinate dimension mismatch");
    if (ncols <= 0) return ISK_OK;
    HIPCHK(hipSetDevice(e->cfg.device));
    const bool is_coords = e->cfg.featurizer != ISK_FEAT_IDENTITY;
    if (is_coords && !e->d_packed) return fail(ISK_EUNSUPPORTED, "pullback needs the register featurizer (<= 65535 atoms)");
    // (vjp_back_kernel keeps delta1, d xhat and the coordinate gradient of a sample in LDS: (H1p + Fp + D) floats -- inside its 160 KiB for
    // every shape isk_create accepts: features + coordinates <= 10240 beyond the register featurizer's range, first hidden width <= 4096)
    const int H1p = isk_pad16(e->cfg.widths[1]);
    const TrainShape ts = train_shape(e);
    if (e->layered) ensure_folded(e);            // the layer-by-layer pullback reads the f32 weight images (forward and transposed)
    const int64_t CH = 16384;
    for (int64_t c0 = 0; c0 < ncols; c0 += CH) {
        const int64_t n = std::min<int64_t>(CH, ncols - c0);
        const int64_t npad = ((n + 15) / 16) * 16;
        TmpBuf in, xh, raw, st, z1, d1, part, ct, gr;
        HIPCHK(hipMalloc(&in.p, (size_t)n * D * sizeof(float)));
        HIPCHK(hipMalloc(&xh.p, (size_t)n * e->Fp * sizeof(float)));
        HIPCHK(hipMalloc(&raw.p, (size_t)n * e->Fp * sizeof(float)));
        HIPCHK(hipMalloc(&st.p, (size_t)n * 2 * sizeof(float)));
        {
            int ks = 1, cpb;
            if (e->native) isk_fwd1n_plan((int)n, e->Fp, H1p / 16, &ks, &cpb);
            HIPCHK(hipMalloc(&z1.p, (size_t)ks * z1_rows(n) * H1p * sizeof(float)));
        }
        HIPCHK(hipMalloc(&d1.p, (size_t)npad * H1p * sizeof(float)));
        if (!e->layered) HIPCHK(hipMalloc(&part.p, (size_t)isk_train_part_floats(ts, (int)n) * sizeof(float)));
        HIPCHK(hipMalloc(&ct.p, (size_t)n * e->nout * sizeof(float)));
        HIPCHK(hipMalloc(&gr.p, (size_t)n * D * sizeof(float)));
        HIPCHK(hipMemcpyAsync(in.p, coords + c0 * D, (size_t)n * D * sizeof(float), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(ct.p, cot + c0 * e->nout, (size_t)n * e->nout * sizeof(float), hipMemcpyHostToDevice, e->stream));
        FeatArgs a = {};
        a.in = (const float*)in.p; a.pairs = is_coords ? e->d_pairs : nullptr; a.packed_pairs = is_coords ? e->d_packed : nullptr;
        a.xhat = (float*)xh.p; a.S = n; a.D = (int)D; a.F = e->F; a.Fp = e->Fp;
        a.normalize = e->cfg.layernorm; a.ln_mode = e->cfg.ln_mode; a.eps = e->cfg.ln_eps;
        a.raw = (float*)raw.p; a.stats = (float*)st.p;
        isk_launch_featurize(a, e->stream);
        const float* delta1 = (const float*)d1.p;
        TmpBuf lz[ISK_MAX_LAYERS], la[ISK_MAX_LAYERS], ldl[ISK_MAX_LAYERS];
        if (e->layered) {
            // layer by layer with the row-GEMM kernels: forward, cotangent seed at the last layer, delta back to layer 0
            const int L = e->L;
            hipStream_t st = e->stream;
            for (int l = 0; l < L; ++l) {
                const size_t w = (size_t)npad * isk_pad16(e->cfg.widths[l + 1]) * sizeof(float);
                HIPCHK(hipMalloc(&lz[l].p, w));
                HIPCHK(hipMalloc(&la[l].p, w));
                HIPCHK(hipMalloc(&ldl[l].p, w));
            }
            for (int l = 0; l < L; ++l) {
                const int N = e->cfg.widths[l + 1], Np = isk_pad16(N);
                const float* X = (l == 0) ? (const float*)xh.p : (const float*)la[l - 1].p;
                const int ldx = (l == 0) ? e->Fp : isk_pad16(e->cfg.widths[l]);
                isk_launch_gemm_rows(X, ldx, e->d_Wt[l], e->d_bias[l], Np / 16, nullptr, (int)n, (float*)lz[l].p, st);
                if (l < L - 1) isk_launch_act((const float*)lz[l].p, (float*)la[l].p, npad, Np, N, act_of_layer(e, l), st);
            }
            const int nop = isk_pad16(e->nout);
            isk_launch_cot_delta((const float*)lz[L - 1].p, nop, (const float*)ct.p, n, e->nout, e->cfg.lastactivation, (float*)ldl[L - 1].p, nop, st);
            for (int l = L - 1; l >= 1; --l) {
                const int K = e->cfg.widths[l], Np = isk_pad16(e->cfg.widths[l + 1]);
                isk_launch_gemm_rows((const float*)ldl[l].p, Np, e->d_WtT[l], e->d_zero, isk_pad16(K) / 16, nullptr, (int)n, (float*)la[l - 1].p, st);
                isk_launch_dact_mul((const float*)la[l - 1].p, (const float*)lz[l - 1].p, (float*)ldl[l - 1].p, npad, isk_pad16(K), K,
                                    act_of_layer(e, l - 1), st);
            }
            delta1 = (const float*)ldl[0].p;
        } else {
            TrainBuffers tb = train_buffers(e);
            tb.xhat_xs = (const float*)xh.p; tb.target = (const float*)ct.p; tb.a1 = (float*)z1.p; tb.delta1 = (float*)d1.p;
            tb.part = (float*)part.p; tb.cot_mode = 1;
            int cpb;
            isk_fwd1n_plan((int)n, e->Fp, H1p / 16, &tb.KS, &cpb);
            tb.zstride = z1_rows(n) * H1p;
            isk_launch_train_fwd1n(ts, tb, nullptr, (int)n, e->stream);
            isk_launch_train_tail(ts, tb, nullptr, (int)n, 1.0f, e->stream);
        }
        VjpArgs v = {};
        v.delta1 = delta1; v.xhat = (const float*)xh.p; v.raw = (const float*)raw.p; v.stats = (const float*)st.p;
        v.coords = (const float*)in.p; v.pairs = is_coords ? e->d_pairs : nullptr; v.theta = e->d_theta;
        v.off_W0 = e->off_W[0]; v.off_lns = e->cfg.layernorm ? e->off_lns : -1;
        v.grad = (float*)gr.p; v.S = n; v.D = (int)D; v.F = e->F; v.Fp = e->Fp; v.H1 = e->cfg.widths[1]; v.H1p = H1p;
        v.ln_mode = e->cfg.ln_mode;
        v.native = e->native;
        isk_launch_vjp_back(v, e->stream);
        HIPCHK(hipMemcpyAsync(grad + c0 * D, gr.p, (size_t)n * D * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        HIPCHK(hipGetLastError());
    }
    return ISK_OK;
}
