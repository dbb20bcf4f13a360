// Data-parallel run over N, host-driven hooks: the caller brings the collectives (torch.distributed over gloo or RCCL,
// MPI.jl) and calls the engine between them.  The native run -- RCCL called by the engine itself -- is engine_dp_native.hip.
// The reference has no multi-device code.
#include "engine.hpp"

// ---------------------------------------------------------------------------------------
// data-parallel hooks
// ---------------------------------------------------------------------------------------
extern "C" int isk_dp_koopman_local(isk_engine* e) {
    CHK(need_data(e));
    if (e->cfg.featurize_every_iteration) featurize_resident(e);
    CHK(forward_resident(e, e->cfg.target_kind == ISK_TARGET_ISA ? 3 : 1, e->cfg.featurize_every_iteration != 0));
    ScopedTimer t(e, ISK_T_TARGET);
    launch_kmean(e);
    return ISK_OK;
}

extern "C" int isk_dp_minmax_local(isk_engine* e, float mm[2]) {
    CHK(need_data(e));
    if (!e->have_ks) return fail(ISK_ESTATE, "call isk_dp_koopman_local first");
    isk_launch_minmax(e->d_ks, e->N * e->nout, e->d_mm, e->stream);
    HIPCHK(hipMemcpyAsync(mm, e->d_mm, 2 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return ISK_OK;
}

extern "C" int isk_dp_shiftscale_apply(isk_engine* e, float gmin, float gmax) {
    CHK(need_data(e));
    if (!e->have_ks) return fail(ISK_ESTATE, "call isk_dp_koopman_local first");
    if (!(gmax > gmin)) return fail(ISK_ECONSTANT_CHI, isk_status_string(ISK_ECONSTANT_CHI));
    const float mm[2] = {gmin, gmax};
    HIPCHK(hipMemcpyAsync(e->d_mm, mm, sizeof(mm), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    isk_launch_shiftscale(e->d_ks, e->d_mm, e->d_target, e->N, e->d_status, e->stream);
    e->have_target = true;
    return ISK_OK;
}

extern "C" int isk_dp_get_ks_chi(isk_engine* e, float* ks, float* chi) {
    CHK(need_data(e));
    if (!e->have_ks) return fail(ISK_ESTATE, "call isk_dp_koopman_local first");
    const size_t nb = (size_t)e->N * e->nout * sizeof(float);
    if (ks) HIPCHK(hipMemcpyAsync(ks, e->d_ks, nb, hipMemcpyDeviceToHost, e->stream));
    if (chi) HIPCHK(hipMemcpyAsync(chi, e->d_out + e->N * e->K * e->nout, nb, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return ISK_OK;
}

extern "C" int isk_dp_isa_apply(isk_engine* e, const double* At, const int32_t* rowperm) {
    CHK(need_data(e));
    if (!e->have_ks || !At || !rowperm) return fail(ISK_ESTATE, "call isk_dp_koopman_local first / null argument");
    const int d = e->nout;
    HIPCHK(hipMemcpyAsync(e->d_At, At, (size_t)d * d * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_rowperm, rowperm, d * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    isk_launch_isa_apply(e->d_ks, e->d_At, e->d_rowperm, e->d_target, e->N, d, e->stream);
    e->have_target = true;
    return ISK_OK;
}

extern "C" int isk_dp_ks_chi_dev(isk_engine* e, void** ks_dev, void** chi_dev, int64_t* nfloats) {
    CHK(need_data(e));
    if (!e->have_ks || !ks_dev || !chi_dev || !nfloats) return fail(ISK_ESTATE, "call isk_dp_koopman_local first / null argument");
    *ks_dev = e->d_ks;
    *chi_dev = e->d_out + e->N * e->K * e->nout;
    *nfloats = e->N * e->nout;
    return ISK_OK;
}

extern "C" int isk_dp_isa_dev(isk_engine* e, const float* ks_all_dev, const float* chi_all_dev, int64_t n_global, int64_t first) {
    return dp_isa_dev(e, ks_all_dev, nullptr, chi_all_dev, n_global, first);
}
// ks64_all_dev: the gathered K chi as the K-means summed it in double (the engine's own native run), or nullptr (hosts that gather
// the Float32 slices themselves)
int dp_isa_dev(isk_engine* e, const float* ks_all_dev, const double* ks64_all_dev, const float* chi_all_dev, long long n_global, long long first) {
    CHK(need_data(e));
    if (e->cfg.target_kind != ISK_TARGET_ISA) return fail(ISK_EINVAL, "engine is not configured for TransformISA");
    if (e->nout > ISK_ISA_DEV_MAXD) return fail(ISK_EUNSUPPORTED, "the device ISA chain covers up to 8 outputs: gather to the host (isk_dp_get_ks_chi, isk_isa_solve, isk_dp_isa_apply)");
    if ((!ks_all_dev && !ks64_all_dev) || !chi_all_dev || n_global < e->N || first < 0 || first + e->N > n_global) return fail(ISK_EINVAL, "bad gathered arrays");
    const int d = e->nout;
    if (e->isa_work_dp_rows < n_global) {
        HIPCHK(hipStreamSynchronize(e->stream));
        free_dev(e->d_isa_work_dp);
        e->d_isa_work_dp = nullptr;
        HIPCHK(hipMalloc((void**)&e->d_isa_work_dp, (size_t)isk_isa_work_doubles(n_global, d) * sizeof(double)));
        e->isa_work_dp_rows = n_global;
    }
    IsaArgs ia;
    ia.ks = ks_all_dev; ia.ks64 = ks64_all_dev; ia.chi = chi_all_dev; ia.work = e->d_isa_work_dp; ia.target = e->d_target;
    ia.lossw = e->d_lossw; ia.status = e->d_status; ia.N = n_global; ia.d = d;
    ia.permute = e->cfg.isa_permute; ia.whitening = e->cfg.isa_whitening;
    ia.out_first = first; ia.out_count = e->N;
    isk_launch_isa(ia, e->stream);
    e->have_target = true;
    return ISK_OK;
}

extern "C" int isk_dp_check_target(isk_engine* e) {
    CHK(need_data(e));
    return check_status(e);
}

extern "C" int isk_dp_set_loss_weights(isk_engine* e, const float* w) {
    if (!e || !w) return fail(ISK_EINVAL, "null argument");
    HIPCHK(hipMemcpyAsync(e->d_lossw, w, (size_t)e->nout * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return ISK_OK;
}

extern "C" int isk_dp_target_moments(isk_engine* e, double* sum, double* sumsq) {
    CHK(need_data(e));
    if (!e->have_target || !sum || !sumsq) return fail(ISK_ESTATE, "no target / null argument");
    const int d = e->nout;
    std::vector<float> t((size_t)e->N * d);
    HIPCHK(hipMemcpyAsync(t.data(), e->d_target, t.size() * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int i = 0; i < d; ++i) { sum[i] = 0; sumsq[i] = 0; }
    for (long long n = 0; n < e->N; ++n)
        for (int i = 0; i < d; ++i) { const double v = t[n * d + i]; sum[i] += v; sumsq[i] += v * v; }
    return ISK_OK;
}

extern "C" int isk_dp_grad_buffer(isk_engine* e, void** buf, int64_t* n) {
    if (!e || !buf || !n) return fail(ISK_EINVAL, "null argument");
    *buf = e->d_grad;
    *n = e->nparams + 1;
    return ISK_OK;
}

extern "C" int isk_dp_epoch_begin(isk_engine* e, const int32_t* perm, int32_t n) {
    CHK(need_data(e));
    if (!e->have_target) return fail(ISK_ESTATE, "no target");
    if (!perm || n < 1 || n > e->N) return fail(ISK_EINVAL, "bad permutation");
    for (int i = 0; i < n; ++i)
        if (perm[i] < 0 || perm[i] >= e->N) return fail(ISK_EINVAL, "permutation entry out of range");
    HIPCHK(hipMemcpyAsync(e->d_perm, perm, (size_t)n * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIPCHK(zero_loss_acc(e));
    HIPCHK(hipStreamSynchronize(e->stream));                 // perm is caller-owned
    e->dp_perm_len = n;
    featurize_ys_beside_training(e, 4);      // the host-driven step loop leaves the GPU even idler
    return ISK_OK;
}

extern "C" int isk_dp_local_grads(isk_engine* e, int32_t first, int32_t nidx, int64_t global_batch) {
    if (!e || e->N == 0) return fail(ISK_ESTATE, "no data");
    if (first < 0 || nidx < 1 || first + nidx > e->dp_perm_len || global_batch < nidx) return fail(ISK_EINVAL, "bad batch");
    launch_grads(e, e->d_perm + first, nidx, 1.0f / (float)global_batch);
    return ISK_OK;
}

extern "C" int isk_dp_apply(isk_engine* e) {
    if (!e || e->N == 0) return fail(ISK_ESTATE, "no data");
    launch_opt(e, 1.0f, !epoch_lazy_fold(e));     // (the host-driven loop has no "last step": the next forward folds on demand)
    return ISK_OK;
}

extern "C" int isk_dp_epoch_end(isk_engine* e, double* loss_sum) {
    CHK(need_data(e));
    double acc[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(acc, e->d_loss_acc, sizeof(acc), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipGetLastError());
    e->dp_perm_len = 0;
    if (acc[1] != 0.0) return fail(ISK_ECOLLAPSED, isk_status_string(ISK_ECOLLAPSED));
    if (loss_sum) *loss_sum = acc[0];
    return ISK_OK;
}
