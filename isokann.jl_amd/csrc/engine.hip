// Host side of the MI355X ISOKANN engine: owns device memory, sequences the kernels of
// kernels_fwd.hip / kernels_fwd_bf16.hip / kernels_train.hip / kernels_isa.hip / kernels_layered.hip on the engine's
// streams (main, side: the next iteration's featurizer, copy: shuffle orders), implements the C ABI of
// include/isokann_hip.h.  Nothing of an isk_run iteration runs on the host: the Float64 inner-simplex (ISA) target is
// a device kernel chain (the reference pulls chi and ks to the CPU, src/isotarget.jl:81-107; the host twin
// isa_solve_host below serves the host-driven data-parallel hooks and isk_isa_solve).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <rocblas/rocblas.h>
#include "isk_internal.h"

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            return fail(_e == hipErrorOutOfMemory ? ISK_ENOMEM : ISK_EHIP,                                \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                               \
    } while (0)

#define CHK(expr)                      \
    do {                               \
        int _r = (expr);               \
        if (_r != ISK_OK) return _r;   \
    } while (0)

extern "C" const char* isk_last_error(void) { return g_err.c_str(); }

extern "C" const char* isk_status_string(int s) {
    switch (s) {
        case ISK_OK: return "ok";
        case ISK_EINVAL: return "invalid argument";
        case ISK_EHIP: return "HIP error";
        case ISK_ENOMEM: return "out of memory";
        case ISK_ECONSTANT_CHI: return "DomainError: Could not compute the shift-scale. chi function is constant";
        case ISK_ECOLLAPSED: return "DomainError: The ISOKANN model collapsed under training. Try reducing the learning rate or increasing regularization";
        case ISK_ESINGULAR: return "DomainError: Could not compute the simplex transformation. The subspace might be singular/collapsed";
        case ISK_EUNSUPPORTED: return "unsupported configuration";
        case ISK_ESTATE: return "invalid call order";
        default: return "unknown status";
    }
}

extern "C" int isk_abi_version(void) { return ISK_ABI_VERSION; }

extern "C" int isk_device_count(int32_t* count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(ISK_EHIP, hipGetErrorString(e)); }
    *count = n;
    return ISK_OK;
}

extern "C" void isk_config_default(isk_config* c) {
    std::memset(c, 0, sizeof(*c));
    c->abi_version = ISK_ABI_VERSION;
    c->device = 0;
    c->n_layers = 0;
    c->activation = ISK_ACT_SIGMOID;        // pairnet default, src/models.jl:65
    c->lastactivation = ISK_ACT_IDENTITY;
    c->layernorm = 1;
    c->ln_mode = ISK_LN_VAR_EPS2;
    c->ln_eps = 1e-5f;
    c->opt_kind = ISK_OPT_NESTEROV;         // src/iso.jl:18
    c->lr = 1e-3f; c->reg = 1e-4f;          // src/models.jl:12,20
    c->beta1 = 0.9f; c->beta2 = 0.999f; c->opt_eps = 1e-8f; c->rho = 0.9f;
    c->featurizer = ISK_FEAT_IDENTITY;
    c->featurize_every_iteration = 0;
    c->target_kind = ISK_TARGET_SHIFTSCALE;
    c->isa_permute = 1; c->isa_whitening = 0;   // src/isotarget.jl:74-77
    c->minibatch = 100;                     // src/iso.jl:10
    c->seed = 0;
    c->forward_mode = 0;
}

// ---------------------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------------------
struct EventPair { hipEvent_t a, b; int id; };

struct isk_engine {
    isk_config cfg;
    int L = 0, F = 0, Fp = 0, Din = 0, nout = 0;
    // ENGINE layout of the flat parameter / gradient / optimizer-state vectors (what the kernels see).  Fused path
    // (`native`): LayerNorm scale and bias padded to Fp, layer-1 weights fragment-linear [Fp/16][nt][64][4] (zero padded),
    // everything else in Flux order; layered path: the Flux order itself.  See kernels_train.hip.
    long long nparams = 0, off_lns = -1, off_lnb = -1;
    long long off_W[ISK_MAX_LAYERS], off_b[ISK_MAX_LAYERS];
    bool native = false;
    // API layout (Flux trainables order, W column-major (out, in)): isk_get/set_params, optimizer state, isk_batch_grads
    long long napi = 0, aoff_lns = -1, aoff_lnb = -1;
    long long aoff_W[ISK_MAX_LAYERS], aoff_b[ISK_MAX_LAYERS];
    bool wt_dirty = true;            // fused path: theta changed since the forward weight images (d_Wt, d_bias) were built
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // side stream: the (weight-independent) ys featurization of the NEXT iteration runs here,
    // concurrently with the latency-bound training kernels of the current one
    hipStream_t side = nullptr;
    hipStream_t copy = nullptr;       // small H2D staging (the epoch's shuffle order): never queued behind the side-stream featurizer
    hipEvent_t ev_fwd_done = nullptr, ev_ys_ready = nullptr;
    long long side_after_forward = 0;   // isk_run / isk_dp_run: launch the side-stream ys featurization right behind the forward
                                        // kernel (value = optimizer steps of the coming training phase), not after the target kernels
    long long early_rows = 0;           // rows of the next iteration already featurized beside THIS iteration's forward (ISK_FEAT_EARLY_PCT)
    bool ys_ahead = false;       // a side-stream featurization of ys is in flight / complete
    // parameters
    float *d_theta = nullptr, *d_grad = nullptr, *d_m = nullptr, *d_v = nullptr;
    float* d_Wt[ISK_MAX_LAYERS] = {nullptr};
    float* d_bias[ISK_MAX_LAYERS] = {nullptr};
    void* d_W16 = nullptr;           // split-bf16 / split-f16 planes of the folded layer-1 weights (forward_mode 1, 2, 3)
    bool w16_dirty = true;
    // forward_mode 3 (split-f16 x2): the featurizer writes pre-split f16 planes + inverse row scales, the xs rows also in f32
    bool planes = false;             // effective: mode 3 requested AND the register featurizer / fused kernels cover the shape
    unsigned* d_wmax = nullptr;      // [2] weight-tensor maximum of the folded layer-1 image: fold_kernel builds it in one slot and clears the
    int wmax_par = 0, wmax_last = 0; //     other for the next fold (no memset node), fold_f16 reads the slot of the latest fold
    float* d_wscale = nullptr;       // [1] inverse scale of the weight planes
    float* d_xscale = nullptr;       // [(N*K + N)] inverse row scales of d_xhat
    float* d_xscale_alt = nullptr;   //             ... of d_xhat_alt
    float* d_xs32 = nullptr;         // [N][Fp] f32 standardised xs rows (training kernels)
    float* d_xs32_alt = nullptr;     //         the copy the side-stream featurizer writes for the next iteration
    // layered path (networks too wide for the fused kernels): per-layer images and HBM activations
    bool layered = false;
    void* d_W16L[ISK_MAX_LAYERS] = {nullptr};     // bf16 planes per layer (forward_mode 1)
    float* d_WtT[ISK_MAX_LAYERS] = {nullptr};     // fragment-linear W' per layer l >= 1 (delta back-propagation)
    float *d_actA = nullptr, *d_actB = nullptr;   // resident forward ping-pong [(N*K+N)][max pad32(H)]
    float* d_lz[ISK_MAX_LAYERS] = {nullptr};      // training: pre-activations per layer [Bpad][pad16(N_l)]
    float* d_la[ISK_MAX_LAYERS] = {nullptr};      //           activations
    float* d_ld[ISK_MAX_LAYERS] = {nullptr};      //           deltas
    float *d_G0 = nullptr, *d_D0 = nullptr;       // raw layer-0 gradient for the LayerNorm unfold
    float* d_xb = nullptr;                        // rocBLAS training path: the batch rows of layer 0, gathered contiguous
    float* d_W0g = nullptr;                       //                        layer-0 weights with the LayerNorm scale folded in (Flux layout)
    rocblas_handle blas = nullptr;
    float* d_zero = nullptr;                      // zero bias for the transposed GEMMs
    double* d_loss_acc = nullptr;
    int* d_status = nullptr;
    float* d_mm = nullptr;
    int2* d_pairs = nullptr;
    unsigned* d_packed = nullptr;
    double bt1 = 0.9, bt2 = 0.999;
    // device copy of [bt1, bt2]: opt_step derives the Adam bias corrections from it and the fold that follows advances it,
    // so a training step's launch arguments never change and an epoch can be replayed as ONE hipGraph (train_epoch_graph)
    double* d_bt = nullptr;
    bool bt_dev_valid = false;
    // the captured epoch: nb x (train_fwd1n, train_tail, train_grad1n, opt_step).  TWO slots: with per-iteration
    // featurization the xs feature rows the launches read alternate between d_xs32 and d_xs32_alt
    struct TrainGraph {
        hipGraphExec_t exec = nullptr;
        long long bs = 0, nb = 0;
        int flags = -1;
        const float* xs = nullptr;
    } tg[2];
    bool tg_disabled = false;
    hipEvent_t ev_loss = nullptr;        // recorded behind the epoch's loss read-back: the host waits for IT, not for the stream
    // data
    long long N = 0, K = 0, D = 0;
    float *d_in_xs = nullptr, *d_in_ys = nullptr;   // raw coords / raw features: ONE allocation, ys rows then xs rows (d_in_xs points inside)
    float* d_xhat = nullptr;                         // CURRENT [(N*K + N)][Fp] buffer: ys rows first, then xs rows
    float* d_xhat_alt = nullptr;                     // the other buffer of the pair (per-iteration featurize mode)
    float* d_out = nullptr;                          // [(N*K + N)][nout]
    float *d_ks = nullptr, *d_target = nullptr, *d_lossw = nullptr;
    float* d_kw = nullptr;                           // optional Koopman sample weights [N][K] (WeightedSamples)
    bool have_target = false, have_ks = false;
    // training workspace
    float *d_z1 = nullptr, *d_delta1 = nullptr, *d_part = nullptr;
    int* d_perm = nullptr;
    hipEvent_t ev_perm = nullptr;   // the shuffle order of the coming epoch was uploaded on the side stream (isk_run)
    double* h_res = nullptr;     // pinned + mapped: [loss sum, collapsed flag, status word], published by opt_step (d_hres = its device address)
    double* d_hres = nullptr;
    bool loss_acc_zeroed = false; // the K-mean kernel of this iteration already zeroed d_loss_acc
    int* h_perm = nullptr;       // pinned staging for the shuffle order: an async copy from pageable memory waits for the stream
    double* d_At = nullptr;      // host-driven data-parallel ISA: the Float64 simplex inverse (applied in Float64, as on the device path)
    int* d_rowperm = nullptr;
    double* d_isa_work = nullptr;
    // native data-parallel run (isk_dp_run): RCCL communicator + gathered ks / chi
    ncclComm_t comm = nullptr;
    int dp_rank = 0, dp_world = 1;
    long long dp_checked_N = -1;     // shard size for which the equal-shards check has passed on this communicator
    float* d_gks = nullptr;
    float* d_gchi = nullptr;
    long long gather_rows = 0;
    double* d_isa_work_dp = nullptr;
    long long isa_work_dp_rows = 0;
    uint64_t rng[4];
    int dp_perm_len = 0;
    int last_nslab = 1;
    // timers
    bool timers_on = false;
    std::vector<EventPair> ev_live, ev_free;
    double t_ms[ISK_T_COUNT] = {0};
    long long t_cnt[ISK_T_COUNT] = {0};
};

// padded input width of layer l: the feature rows are padded to 32; hidden activations to 16 on the fused path
// and to 32 on the layered path (the split-bf16 GEMM contracts 32 at a time)
static int layer_kp(const isk_engine* e, int l) {
    if (l == 0) return e->Fp;
    return e->layered ? isk_pad32(e->cfg.widths[l]) : isk_pad16(e->cfg.widths[l]);
}

static void free_dev(void* p) { if (p) (void)hipFree(p); }

static void free_data(isk_engine* e) {
    if (e->side) (void)hipStreamSynchronize(e->side);
    e->ys_ahead = false;
    for (auto& g : e->tg) { if (g.exec) (void)hipGraphExecDestroy(g.exec); g = isk_engine::TrainGraph(); }
    free_dev(e->d_in_ys); free_dev(e->d_xs32_alt); e->d_xs32_alt = nullptr; free_dev(e->d_xhat); free_dev(e->d_xhat_alt); free_dev(e->d_out); free_dev(e->d_ks);
    free_dev(e->d_xscale); free_dev(e->d_xscale_alt); free_dev(e->d_xs32);
    e->d_xscale = e->d_xscale_alt = e->d_xs32 = nullptr;
    free_dev(e->d_target); free_dev(e->d_kw); e->d_kw = nullptr; free_dev(e->d_z1); free_dev(e->d_delta1); free_dev(e->d_part); free_dev(e->d_perm); free_dev(e->d_isa_work);
    e->d_isa_work = nullptr;
    free_dev(e->d_actA); free_dev(e->d_actB); e->d_actA = e->d_actB = nullptr;
    free_dev(e->d_xb); e->d_xb = nullptr;
    for (int l = 0; l < ISK_MAX_LAYERS; ++l) { free_dev(e->d_lz[l]); free_dev(e->d_la[l]); free_dev(e->d_ld[l]); e->d_lz[l] = e->d_la[l] = e->d_ld[l] = nullptr; }
    e->d_in_xs = e->d_in_ys = e->d_xhat = e->d_xhat_alt = e->d_out = e->d_ks = e->d_target = e->d_z1 = e->d_delta1 = e->d_part = nullptr;
    e->d_perm = nullptr;
    if (e->h_perm) { (void)hipHostFree(e->h_perm); e->h_perm = nullptr; }
    e->N = e->K = e->D = 0;
    e->have_target = e->have_ks = false;
}

// splitmix64 / xoshiro256** (documented shuffle stream; Julia's cannot be reproduced)
static uint64_t splitmix(uint64_t& x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t xoshiro(uint64_t s[4]) {
    const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
}
static void shuffle_perm(isk_engine* e, std::vector<int32_t>& p) {
    for (size_t i = 0; i < p.size(); ++i) p[i] = (int32_t)i;
    for (size_t i = p.size(); i > 1; --i) {
        const size_t j = (size_t)(xoshiro(e->rng) % i);
        std::swap(p[i - 1], p[j]);
    }
}

// ---- timers ---------------------------------------------------------------------------
struct ScopedTimer {
    isk_engine* e; EventPair ep; bool on; hipStream_t st;
    ScopedTimer(isk_engine* e_, int id, hipStream_t s = nullptr) : e(e_), on(e_->timers_on), st(s ? s : e_->stream) {
        if (!on) return;
        if (!e->ev_free.empty()) { ep = e->ev_free.back(); e->ev_free.pop_back(); }
        else { (void)hipEventCreate(&ep.a); (void)hipEventCreate(&ep.b); }
        ep.id = id;
        (void)hipEventRecord(ep.a, st);
    }
    ~ScopedTimer() {
        if (!on) return;
        (void)hipEventRecord(ep.b, st);
        e->ev_live.push_back(ep);
    }
};

static void drain_timers(isk_engine* e) {
    for (auto& ep : e->ev_live) {
        (void)hipEventSynchronize(ep.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) { e->t_ms[ep.id] += ms; e->t_cnt[ep.id] += 1; }
        e->ev_free.push_back(ep);
    }
    e->ev_live.clear();
}

// ---- argument builders ----------------------------------------------------------------
static int act_of_layer(const isk_engine* e, int l) { return l == e->L - 1 ? e->cfg.lastactivation : e->cfg.activation; }

static MlpFwdArgs fwd_args(const isk_engine* e, const float* X, long long S, float* out) {
    MlpFwdArgs a;
    std::memset(&a, 0, sizeof(a));
    a.X = X; a.rows = nullptr; a.S = S; a.Fp = e->Fp; a.nchunks = e->Fp / 16;
    a.W1t = e->d_Wt[0]; a.bias1 = e->d_bias[0];
    a.act1 = act_of_layer(e, 0);
    a.H1 = e->cfg.widths[1];
    a.nt1 = isk_pad16(a.H1) / 16;
    a.ntail = e->L - 1;
    int sa = isk_pad16(e->cfg.widths[1]) + 4, sb = 0;    // layer-1 output always passes through scratch A
    for (int l = 1; l < e->L; ++l) {
        TailLayer& t = a.tail[l - 1];
        t.Wt = e->d_Wt[l]; t.bias = e->d_bias[l];
        t.kchunks = isk_pad16(e->cfg.widths[l]) / 16;
        t.ntiles = isk_pad16(e->cfg.widths[l + 1]) / 16;
        t.act = act_of_layer(e, l);
        t.width = e->cfg.widths[l + 1];
        t.lds_off = a.tailw_floats;
        a.tailw_floats += t.kchunks * t.ntiles * 256;
    }
    // scratch A holds the outputs of layers 0,2,4,..; scratch B of layers 1,3,..
    for (int l = 0; l < e->L - 1; ++l) {
        const int w = isk_pad16(e->cfg.widths[l + 1]) + 4;
        if (l % 2 == 0) sa = std::max(sa, w); else sb = std::max(sb, w);
    }
    a.stride_a = std::max(sa, 4); a.stride_b = std::max(sb, 4);
    a.out = out; a.nout = e->nout;
    a.wscale = e->d_wscale;          // a.xscale is set by the caller that owns the plane rows (forward_mode 3)
    isk_fwd_layout_biases(a);
    static const int skew = [] { const char* v = getenv("ISK_BF16_SKEW"); return v ? atoi(v) : 1; }();
    a.sched_skew = skew;
    return a;
}

static TrainShape train_shape(const isk_engine* e) {
    TrainShape s;
    std::memset(&s, 0, sizeof(s));
    s.nlayers = e->L;
    for (int i = 0; i <= e->L; ++i) s.widths[i] = e->cfg.widths[i];
    s.act = e->cfg.activation; s.lastact = e->cfg.lastactivation;
    s.layernorm = e->cfg.layernorm;
    s.off_lns = e->off_lns; s.off_lnb = e->off_lnb;
    for (int l = 0; l < e->L; ++l) { s.off_W[l] = e->off_W[l]; s.off_b[l] = e->off_b[l]; }
    s.nparams = e->nparams;
    return s;
}

static long long z1_rows(long long B) { return ((B + 31) / 32) * 32; }
static TrainBuffers train_buffers(const isk_engine* e) {
    TrainBuffers b;
    std::memset(&b, 0, sizeof(b));
    b.beta1 = e->cfg.beta1; b.beta2 = e->cfg.beta2;
    b.xhat_xs = e->d_xs32;      // a buffer of its own: d_xhat / d_xhat_alt swap every iteration (and hold f16 planes in forward_mode 3)
    b.target = e->d_target; b.lossw = e->d_lossw; b.theta = e->d_theta;
    b.W1t = e->d_Wt[0]; b.bias1 = e->d_bias[0];
    b.a1 = e->d_z1; b.delta1 = e->d_delta1; b.part = e->d_part; b.grad = e->d_grad;
    b.loss_acc = e->d_loss_acc; b.Fp = e->Fp;
    return b;
}

static void launch_fold(isk_engine* e) {
    FoldArgs a;
    std::memset(&a, 0, sizeof(a));
    a.native = e->native;
    if (e->planes) {
        a.wmax = e->d_wmax + e->wmax_par; a.wmax_clear = e->d_wmax + (e->wmax_par ^ 1);
        e->wmax_last = e->wmax_par; e->wmax_par ^= 1;
    }
    a.theta = e->d_theta; a.nlayers = e->L;
    for (int l = 0; l < e->L; ++l) a.Kp[l] = layer_kp(e, l);
    for (int i = 0; i <= e->L; ++i) a.widths[i] = e->cfg.widths[i];
    a.off_lns = e->off_lns; a.off_lnb = e->off_lnb;
    for (int l = 0; l < e->L; ++l) { a.off_W[l] = e->off_W[l]; a.off_b[l] = e->off_b[l]; a.Wt[l] = e->d_Wt[l]; a.bias[l] = e->d_bias[l]; }
    isk_launch_fold(a, e->stream);
    if (e->layered)
        for (int l = 1; l < e->L; ++l)
            isk_launch_fold_T(e->d_theta, e->off_W[l], e->cfg.widths[l], e->cfg.widths[l + 1], e->d_WtT[l], e->stream);
    e->w16_dirty = true;
    e->wt_dirty = false;
}
// fused path: the training steps read theta itself, the forward weight images are rebuilt when a forward needs them
static void ensure_folded(isk_engine* e) {
    if (e->wt_dirty) launch_fold(e);
}

// ---- API (Flux) order <-> engine order, on the host ------------------------------------
static inline long long w1_native_index(const isk_engine* e, int o, int k) {
    const int nt = isk_pad16(e->cfg.widths[1]) / 16;
    return e->off_W[0] + (((long long)(k >> 4) * nt + (o >> 4)) * 64 + (((k & 15) >> 2) << 4 | (o & 15))) * 4 + (k & 3);
}
static void api_to_engine(const isk_engine* e, const float* a, std::vector<float>& out) {
    out.assign((size_t)e->nparams, 0.f);
    if (!e->native) { std::memcpy(out.data(), a, (size_t)e->napi * sizeof(float)); return; }
    const int F = e->F, H1 = e->cfg.widths[1];
    if (e->cfg.layernorm) {
        std::memcpy(out.data() + e->off_lns, a + e->aoff_lns, (size_t)F * sizeof(float));
        std::memcpy(out.data() + e->off_lnb, a + e->aoff_lnb, (size_t)F * sizeof(float));
    }
    for (int k = 0; k < F; ++k)
        for (int o = 0; o < H1; ++o) out[w1_native_index(e, o, k)] = a[e->aoff_W[0] + o + (long long)k * H1];
    std::memcpy(out.data() + e->off_b[0], a + e->aoff_b[0], (size_t)(e->napi - e->aoff_b[0]) * sizeof(float));   // b1 and every later block
}
static void engine_to_api(const isk_engine* e, const float* in, float* a) {
    if (!e->native) { std::memcpy(a, in, (size_t)e->napi * sizeof(float)); return; }
    const int F = e->F, H1 = e->cfg.widths[1];
    if (e->cfg.layernorm) {
        std::memcpy(a + e->aoff_lns, in + e->off_lns, (size_t)F * sizeof(float));
        std::memcpy(a + e->aoff_lnb, in + e->off_lnb, (size_t)F * sizeof(float));
    }
    for (int k = 0; k < F; ++k)
        for (int o = 0; o < H1; ++o) a[e->aoff_W[0] + o + (long long)k * H1] = in[w1_native_index(e, o, k)];
    std::memcpy(a + e->aoff_b[0], in + e->off_b[0], (size_t)(e->napi - e->aoff_b[0]) * sizeof(float));
}

// ---------------------------------------------------------------------------------------
// lifetime
// ---------------------------------------------------------------------------------------
extern "C" int isk_create(const isk_config* cfg, isk_engine** out) {
    if (!cfg || !out) return fail(ISK_EINVAL, "null argument");
    *out = nullptr;
    if (cfg->abi_version != ISK_ABI_VERSION) return fail(ISK_EINVAL, "abi_version mismatch");
    if (cfg->n_layers < 1 || cfg->n_layers > ISK_MAX_LAYERS) return fail(ISK_EINVAL, "n_layers out of range");
    for (int i = 0; i <= cfg->n_layers; ++i)
        if (cfg->widths[i] < 1) return fail(ISK_EINVAL, "layer width < 1");
    if (cfg->activation < 0 || cfg->activation > ISK_ACT_SWISH || cfg->lastactivation < 0 || cfg->lastactivation > ISK_ACT_SWISH)
        return fail(ISK_EINVAL, "unknown activation");
    if (cfg->opt_kind != ISK_OPT_ADAM && cfg->opt_kind != ISK_OPT_NESTEROV) return fail(ISK_EINVAL, "unknown optimizer");
    const int nout = cfg->widths[cfg->n_layers];
    if (cfg->target_kind < ISK_TARGET_SHIFTSCALE || cfg->target_kind > ISK_TARGET_HOST) return fail(ISK_EINVAL, "unknown target kind");
    if (cfg->target_kind == ISK_TARGET_SHIFTSCALE && nout != 1)
        return fail(ISK_EINVAL, "TransformShiftscale only works with one dimensional chi functions");
    if (cfg->target_kind == ISK_TARGET_ISA && nout < 2)
        return fail(ISK_EINVAL, "TransformISA does not work with one dimensional chi functions");
    if (cfg->target_kind == ISK_TARGET_ISA && nout > 8) return fail(ISK_EUNSUPPORTED, "ISA fixperm is brute force over d! permutations; d <= 8");
    for (int l = 1; l <= cfg->n_layers; ++l)
        if (cfg->widths[l] > 4096) return fail(ISK_EUNSUPPORTED, "layer width > 4096");
    int F = cfg->widths[0], Din = F;
    std::vector<int2> pairs;
    if (cfg->featurizer == ISK_FEAT_ALLPAIRS) {
        const int A = cfg->n_atoms;
        if (A < 2) return fail(ISK_EINVAL, "n_atoms < 2");
        if ((long long)A * (A - 1) / 2 != F) return fail(ISK_EINVAL, "widths[0] != n_atoms*(n_atoms-1)/2");
        // halfinds order (src/utils/pairdists.jl:50-56): for j in 2:n, for i in 1:j-1
        for (int j = 1; j < A; ++j)
            for (int i = 0; i < j; ++i) pairs.push_back(make_int2(i, j));
        Din = 3 * A;
    } else if (cfg->featurizer == ISK_FEAT_PAIRLIST) {
        if (cfg->n_pairs != F || !cfg->pairs) return fail(ISK_EINVAL, "n_pairs must equal widths[0] and pairs be non-null");
        for (int i = 0; i < F; ++i) {
            const int a = cfg->pairs[2 * i], b = cfg->pairs[2 * i + 1];
            if (a < 0 || b < 0 || a >= cfg->n_atoms || b >= cfg->n_atoms) return fail(ISK_EINVAL, "pair index out of range");
            pairs.push_back(make_int2(a, b));
        }
        Din = 3 * cfg->n_atoms;
    } else if (cfg->featurizer != ISK_FEAT_IDENTITY) {
        return fail(ISK_EINVAL, "unknown featurizer");
    }
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(ISK_EINVAL, "device ordinal out of range");
    HIPCHK(hipSetDevice(cfg->device));

    isk_engine* e = new isk_engine();
    e->cfg = *cfg;
    e->cfg.pairs = nullptr;
    e->L = cfg->n_layers; e->F = F; e->Fp = isk_pad32(F); e->Din = Din; e->nout = nout;
    if (const char* fm = getenv("ISK_FWD_MODE")) e->cfg.forward_mode = atoi(fm);
    if (e->cfg.forward_mode < 0 || e->cfg.forward_mode > 3) { delete e; return fail(ISK_EINVAL, "unknown forward_mode"); }
    {
        bool fits = isk_pad16(cfg->widths[1]) / 16 <= ISK_MAX_NT;
        size_t tailw = 0, tail_train = 0;
        int sa = isk_pad16(cfg->widths[1]) + 4, sb = 4;
        for (int l = 1; l < e->L; ++l) {
            tailw += (size_t)isk_pad16(cfg->widths[l]) * isk_pad16(cfg->widths[l + 1]);
            tail_train += (size_t)cfg->widths[l] * cfg->widths[l + 1];
            const int w = isk_pad16(cfg->widths[l + 1]) + 4;
            if (l + 1 < e->L) { if (l % 2 == 0) sa = std::max(sa, w); else sb = std::max(sb, w); }
        }
        for (int l = 0; l < e->L; ++l) tail_train += (size_t)33 * cfg->widths[l + 1];   // 16 x (A, Z) per layer + the bias
        if ((tailw + (size_t)8 * 16 * (sa + sb)) * sizeof(float) > 150 * 1024) fits = false;   // fused forward LDS
        if ((tail_train + 8 * 256) * sizeof(float) > 150 * 1024) fits = false;                  // train_tail LDS (+ its partial-tile buffer)
        if (getenv("ISK_FORCE_LAYERED")) fits = false;
        e->layered = !fits;
    }
    long long off = 0;
    if (cfg->layernorm) { e->aoff_lns = off; off += F; e->aoff_lnb = off; off += F; }
    for (int l = 0; l < e->L; ++l) {
        e->aoff_W[l] = off; off += (long long)cfg->widths[l] * cfg->widths[l + 1];
        e->aoff_b[l] = off; off += cfg->widths[l + 1];
    }
    e->napi = off;
    e->native = !e->layered;
    off = 0;
    const long long lnlen = e->native ? e->Fp : F;
    if (cfg->layernorm) { e->off_lns = off; off += lnlen; e->off_lnb = off; off += lnlen; }
    for (int l = 0; l < e->L; ++l) {
        e->off_W[l] = off;
        off += (l == 0 && e->native) ? (long long)e->Fp * isk_pad16(cfg->widths[1]) : (long long)cfg->widths[l] * cfg->widths[l + 1];
        e->off_b[l] = off; off += cfg->widths[l + 1];
    }
    e->nparams = off;
    uint64_t sm = cfg->seed ^ 0x1234567887654321ull;
    for (int i = 0; i < 4; ++i) e->rng[i] = splitmix(sm);
    e->bt1 = cfg->beta1; e->bt2 = cfg->beta2;

#define ALLOC(ptr, bytes)                                                          \
    do {                                                                           \
        hipError_t _e = hipMalloc((void**)&(ptr), (bytes));                        \
        if (_e != hipSuccess) { isk_destroy(e); return fail(ISK_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(_e)); } \
    } while (0)
    hipError_t se = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (se != hipSuccess) { delete e; return fail(ISK_EHIP, hipGetErrorString(se)); }
    e->own_stream = true;
    if (hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&e->copy, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_fwd_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_ys_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_perm, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_loss, hipEventDisableTiming) != hipSuccess) {
        isk_destroy(e);
        return fail(ISK_EHIP, "side stream / event creation failed");
    }
    if (hipHostMalloc((void**)&e->h_res, 4 * sizeof(double), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&e->d_hres, e->h_res, 0) != hipSuccess) { isk_destroy(e); return fail(ISK_ENOMEM, "hipHostMalloc (mapped)"); }
    const size_t pb = (size_t)(e->nparams + 2) * sizeof(float);
    ALLOC(e->d_theta, pb); ALLOC(e->d_grad, pb * ISK_MAX_SLABS); ALLOC(e->d_m, pb); ALLOC(e->d_v, pb);
    ALLOC(e->d_loss_acc, 4 * sizeof(double)); ALLOC(e->d_mm, 2 * sizeof(float));
    e->d_status = (int*)(e->d_loss_acc + 2);         // the target's status word lives behind the loss accumulator (opt_step publishes both)
    ALLOC(e->d_bt, 4 * sizeof(double));
    ALLOC(e->d_wmax, 2 * sizeof(unsigned)); ALLOC(e->d_wscale, sizeof(float));
    (void)hipMemset(e->d_wmax, 0, 2 * sizeof(unsigned));
    ALLOC(e->d_lossw, (size_t)nout * sizeof(float));
    ALLOC(e->d_At, (size_t)nout * nout * sizeof(double)); ALLOC(e->d_rowperm, (size_t)nout * sizeof(int));
    ALLOC(e->d_W16, (size_t)isk_bf16_chunks(e->Fp) * 32 * isk_pad16(cfg->widths[1]) * 3 * sizeof(uint16_t));
    for (int l = 0; l < e->L; ++l) {
        ALLOC(e->d_Wt[l], (size_t)layer_kp(e, l) * isk_pad16(cfg->widths[l + 1]) * sizeof(float));
        ALLOC(e->d_bias[l], (size_t)isk_pad16(cfg->widths[l + 1]) * sizeof(float));
    }
    if (!pairs.empty()) {
        ALLOC(e->d_pairs, pairs.size() * sizeof(int2));
        (void)hipMemcpy(e->d_pairs, pairs.data(), pairs.size() * sizeof(int2), hipMemcpyHostToDevice);
        if (cfg->n_atoms <= 65535) {
            std::vector<unsigned> pk(pairs.size());
            for (size_t i = 0; i < pairs.size(); ++i) pk[i] = (unsigned)pairs[i].x | ((unsigned)pairs[i].y << 16);
            ALLOC(e->d_packed, pk.size() * sizeof(unsigned));
            (void)hipMemcpy(e->d_packed, pk.data(), pk.size() * sizeof(unsigned), hipMemcpyHostToDevice);
        }
    }
    // fused kernels keep the whole tail in LDS; anything wider runs layer by layer through HBM
    if (e->layered) {
        for (int l = 1; l < e->L; ++l) {
            ALLOC(e->d_WtT[l], (size_t)isk_pad16(cfg->widths[l]) * isk_pad16(cfg->widths[l + 1]) * sizeof(float));
        }
        for (int l = 0; l < e->L; ++l)
            ALLOC(e->d_W16L[l], (size_t)isk_bf16_chunks(layer_kp(e, l)) * 32 * isk_pad16(cfg->widths[l + 1]) * 3 * sizeof(uint16_t));
        ALLOC(e->d_G0, (size_t)F * cfg->widths[1] * sizeof(float));
        ALLOC(e->d_D0, (size_t)cfg->widths[1] * sizeof(float));
        ALLOC(e->d_W0g, (size_t)F * cfg->widths[1] * sizeof(float));
        ALLOC(e->d_zero, (size_t)4096 * sizeof(float));
        (void)hipMemset(e->d_zero, 0, (size_t)4096 * sizeof(float));
    }
#undef ALLOC
    if (e->cfg.forward_mode == 3) {
        // split-f16 needs the register featurizer (it writes the planes) and the fused forward kernel; anything else
        // (wide networks on the layered path, > 2048 features, > 682 atoms) runs the split-bf16 forward instead
        FeatArgs fa = {};
        fa.pairs = e->d_pairs; fa.packed_pairs = e->d_packed; fa.D = e->Din; fa.F = e->F; fa.Fp = e->Fp;
        MlpFwdArgs f = fwd_args(e, nullptr, 0, nullptr);
        f.xscale = e->d_wscale;      // any non-null pointer: only the shape is inspected
        e->planes = !e->layered && isk_featurize_reg_ok(fa) && isk_mlp_fwd_f16x2_fits(f);
        if (!e->planes) e->cfg.forward_mode = 1;
    }
    (void)hipMemset(e->d_theta, 0, pb); (void)hipMemset(e->d_grad, 0, pb * ISK_MAX_SLABS);
    (void)hipMemset(e->d_m, 0, pb); (void)hipMemset(e->d_v, 0, pb);
    (void)hipMemset(e->d_loss_acc, 0, 4 * sizeof(double));
    std::vector<float> ones(nout, 1.0f);
    (void)hipMemcpy(e->d_lossw, ones.data(), nout * sizeof(float), hipMemcpyHostToDevice);
    launch_fold(e);
    (void)hipStreamSynchronize(e->stream);
    *out = e;
    return ISK_OK;
}

static void dp_comm_release(isk_engine* e);
static void blas_release(isk_engine* e);

extern "C" void isk_destroy(isk_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->cfg.device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->side) (void)hipStreamSynchronize(e->side);
    if (e->copy) (void)hipStreamSynchronize(e->copy);
    free_data(e);
    free_dev(e->d_theta); free_dev(e->d_grad); free_dev(e->d_m); free_dev(e->d_v); free_dev(e->d_bt);
    free_dev(e->d_loss_acc); free_dev(e->d_mm); free_dev(e->d_lossw);
    free_dev(e->d_At); free_dev(e->d_rowperm); free_dev(e->d_pairs); free_dev(e->d_packed); free_dev(e->d_isa_work_dp);
    for (int l = 0; l < ISK_MAX_LAYERS; ++l) { free_dev(e->d_Wt[l]); free_dev(e->d_bias[l]); }
    free_dev(e->d_W16); free_dev(e->d_wmax); free_dev(e->d_wscale);
    for (int l = 0; l < ISK_MAX_LAYERS; ++l) { free_dev(e->d_W16L[l]); free_dev(e->d_WtT[l]); }
    free_dev(e->d_G0); free_dev(e->d_D0); free_dev(e->d_zero); free_dev(e->d_W0g);
    blas_release(e);
    dp_comm_release(e);
    free_dev(e->d_gks); free_dev(e->d_gchi);
    drain_timers(e);
    for (auto& ep : e->ev_free) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    if (e->ev_fwd_done) (void)hipEventDestroy(e->ev_fwd_done);
    if (e->ev_ys_ready) (void)hipEventDestroy(e->ev_ys_ready);
    if (e->ev_perm) (void)hipEventDestroy(e->ev_perm);
    if (e->ev_loss) (void)hipEventDestroy(e->ev_loss);
    if (e->h_res) (void)hipHostFree(e->h_res);
    if (e->side) (void)hipStreamDestroy(e->side);
    if (e->copy) (void)hipStreamDestroy(e->copy);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

extern "C" int isk_synchronize(isk_engine* e) {
    if (!e) return fail(ISK_EINVAL, "null engine");
    HIPCHK(hipStreamSynchronize(e->stream));
    return ISK_OK;
}

extern "C" int isk_set_stream(isk_engine* e, void* s) {
    if (!e) return fail(ISK_EINVAL, "null engine");
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->own_stream) { (void)hipStreamDestroy(e->stream); e->own_stream = false; }
    for (auto& g : e->tg) { if (g.exec) (void)hipGraphExecDestroy(g.exec); g = isk_engine::TrainGraph(); }
    e->stream = (hipStream_t)s;
    return ISK_OK;
}

// ---------------------------------------------------------------------------------------
// parameters / optimizer state
// ---------------------------------------------------------------------------------------
extern "C" int isk_nparams(isk_engine* e, int64_t* n) {
    if (!e || !n) return fail(ISK_EINVAL, "null argument");
    *n = e->napi;
    return ISK_OK;
}

// flat vectors cross the ABI in Flux order; the engine order (fragment-linear layer-1 weights on the fused path) stays inside
static int upload_flat(isk_engine* e, float* dst, const float* api) {
    std::vector<float> v;
    api_to_engine(e, api, v);
    HIPCHK(hipMemcpyAsync(dst, v.data(), (size_t)e->nparams * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));          // `v` is pageable and dies here
    return ISK_OK;
}
static int download_flat(isk_engine* e, const float* src, float* api) {
    std::vector<float> v((size_t)e->nparams);
    HIPCHK(hipMemcpyAsync(v.data(), src, (size_t)e->nparams * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    engine_to_api(e, v.data(), api);
    return ISK_OK;
}

extern "C" int isk_set_params(isk_engine* e, const float* flat, int64_t n) {
    if (!e || !flat) return fail(ISK_EINVAL, "null argument");
    if (n != e->napi) return fail(ISK_EINVAL, "parameter count mismatch");
    HIPCHK(hipSetDevice(e->cfg.device));
    CHK(upload_flat(e, e->d_theta, flat));
    launch_fold(e);
    HIPCHK(hipStreamSynchronize(e->stream));
    e->have_target = false; e->have_ks = false;
    return ISK_OK;
}

extern "C" int isk_get_params(isk_engine* e, float* flat, int64_t n) {
    if (!e || !flat) return fail(ISK_EINVAL, "null argument");
    if (n != e->napi) return fail(ISK_EINVAL, "parameter count mismatch");
    return download_flat(e, e->d_theta, flat);
}

extern "C" int isk_set_opt_state(isk_engine* e, const float* m, const float* v, int64_t n, const double bt[2]) {
    if (!e || !m) return fail(ISK_EINVAL, "null argument");
    if (n != e->napi) return fail(ISK_EINVAL, "parameter count mismatch");
    CHK(upload_flat(e, e->d_m, m));
    if (v) CHK(upload_flat(e, e->d_v, v));
    if (bt) { e->bt1 = bt[0]; e->bt2 = bt[1]; e->bt_dev_valid = false; }
    return ISK_OK;
}

extern "C" int isk_get_opt_state(isk_engine* e, float* m, float* v, int64_t n, double bt[2]) {
    if (!e) return fail(ISK_EINVAL, "null argument");
    if (n != e->napi) return fail(ISK_EINVAL, "parameter count mismatch");
    if (m) CHK(download_flat(e, e->d_m, m));
    if (v) CHK(download_flat(e, e->d_v, v));
    if (bt) { bt[0] = e->bt1; bt[1] = e->bt2; }
    return ISK_OK;
}

extern "C" int isk_reset_opt_state(isk_engine* e) {
    if (!e) return fail(ISK_EINVAL, "null engine");
    HIPCHK(hipMemsetAsync(e->d_m, 0, (size_t)e->nparams * sizeof(float), e->stream));
    HIPCHK(hipMemsetAsync(e->d_v, 0, (size_t)e->nparams * sizeof(float), e->stream));
    e->bt1 = e->cfg.beta1; e->bt2 = e->cfg.beta2; e->bt_dev_valid = false;
    return ISK_OK;
}

// ---------------------------------------------------------------------------------------
// data
// ---------------------------------------------------------------------------------------
// Featurizes S rows into rows [row0, row0 + S) of the feature buffer `xhat` (its BASE).  xscale != nullptr: split-f16 planes
// (forward_mode 3; unit-blocked rows, isk_plane_off) with the inverse row scales at xscale[row0 ..]; xs32: the rows from
// local row x32_first on in f32 as well
static void featurize_rows(isk_engine* e, const float* in, long long S, float* xhat, long long row0, hipStream_t st = nullptr,
                           float* xscale = nullptr, float* xs32 = nullptr, long long x32_first = 0, int bg_wgs = 1) {
    FeatArgs a = {};
    a.in = in; a.pairs = e->d_pairs; a.packed_pairs = e->d_packed; a.S = S; a.D = e->Din; a.F = e->F; a.Fp = e->Fp;
    a.normalize = e->cfg.layernorm; a.ln_mode = e->cfg.ln_mode; a.eps = e->cfg.ln_eps;
    a.background = (st != nullptr) ? bg_wgs : 0;       // 1: the default throttle; > 1: that many workgroups in total
    a.planes = xscale != nullptr;
    if (a.planes) { a.xhat = xhat; a.row0 = row0; a.xscale = xscale + row0; }
    else a.xhat = xhat + row0 * e->Fp;
    a.xhat32 = xs32; a.x32_first = x32_first;
    isk_launch_featurize(a, st ? st : e->stream);
}

// per-iteration featurization.  ys (the 95 % share) may already have been produced on the side
// stream, into the other buffer of the pair, while the previous iteration's MFMA-bound forward was
// running; then the main stream only waits for it and swaps buffers.
static void fold_ahead(isk_engine* e);
static void featurize_resident(isk_engine* e) {
    if (e->ys_ahead) {
        // the xs rows of the other buffer do not depend on the side stream: featurize them BEFORE waiting for it, so
        // the (un-throttled) xs launch runs under the tail of the throttled ys featurization instead of after it
        // the side-stream launch covered the ys AND the xs rows (one launch over the contiguous input slab), the f32 xs
        // copy for the training kernels included: swap the three buffer pairs
        static const bool dbg_copy = [] { const char* v = getenv("ISK_DEBUG_FEAT_COPY"); return v && atoi(v) != 0; }();
        if (!dbg_copy) {                 // (diagnostic stream instead of the featurizer: the other buffers hold no features)
            std::swap(e->d_xhat, e->d_xhat_alt);
            std::swap(e->d_xscale, e->d_xscale_alt);
            std::swap(e->d_xs32, e->d_xs32_alt);
        }
        e->ys_ahead = false;
        fold_ahead(e);                  // the weight planes of the coming forward (normally already built behind the last epoch)
        (void)hipStreamWaitEvent(e->stream, e->ev_ys_ready, 0);
        return;
    } else {
        ScopedTimer t(e, ISK_T_FEATURIZE);
        featurize_rows(e, e->d_in_ys, e->N * e->K, e->d_xhat, 0, nullptr, e->d_xscale);
    }
    ScopedTimer t(e, ISK_T_FEATURIZE);
    featurize_rows(e, e->d_in_xs, e->N, e->d_xhat, e->N * e->K, nullptr, e->d_xscale, e->d_xs32);
}

// Where the next iteration's ys featurization runs (ISK_OVERLAP_FEATURIZE):
//   2 (default)  on the side stream BESIDE THE TRAINING KERNELS of this iteration, throttled to one
//                (software-pipelined) workgroup per CU: the training steps are latency-bound chains of small launches that
//                leave most CUs idle, the featurizer is HBM-bound.  Measured on C3: 1.90 -> 1.74 ms/step.
//   1            beside the MFMA-bound forward: measured net LOSS (forward 0.57 -> 0.75 ms), kept for tuning
//   0            inline on the main stream
static int overlap_mode() {
    static const int mode = [] { const char* v = getenv("ISK_OVERLAP_FEATURIZE"); return v ? atoi(v) : 2; }();
    return mode;
}

// Called right after this iteration's forward has been enqueued on the main stream: featurize ys for
// the NEXT iteration on the side stream into the other buffer.  The featurizer is HBM-bound, the
// forward f32-MFMA-bound, so the two share the chip.  `ev_main_point` (recorded before the forward)
// orders the write after every earlier reader of that buffer.
static void featurize_ys_ahead(isk_engine* e) {
    // Measured on MI355X (round 1): co-running the HBM-bound featurizer with the MFMA-bound forward
    // slows the forward from 0.83 to 1.5 ms (the featurizer's streaming evicts the L2-resident
    // weights and saturates the memory queues), a net loss against running them back to back.
    // Kept behind ISK_OVERLAP_FEATURIZE=1 for further tuning.
    if (overlap_mode() != 1 || !e->d_xhat_alt) return;
    (void)hipStreamWaitEvent(e->side, e->ev_fwd_done, 0);
    {
        ScopedTimer t(e, ISK_T_FEATURIZE, e->side);
        featurize_rows(e, e->d_in_ys, e->N * e->K + e->N, e->d_xhat_alt, 0, e->side, e->d_xscale_alt, e->d_xs32_alt, e->N * e->K);
    }
    (void)hipEventRecord(e->ev_ys_ready, e->side);
    e->ys_ahead = true;
}

// mode 2: called once the target of this iteration is enqueued (every earlier reader of the other
// buffer precedes this point on the main stream)
static void featurize_ys_beside_training(isk_engine* e, long long train_launch_steps) {
    // only worth it when the training phase is a long chain of small launches: with a handful of big steps the
    // throttled featurizer would outlast them and become the critical path (measured at minibatch = 0)
    static const long long min_steps = [] { const char* v = getenv("ISK_SIDE_MIN_STEPS"); return v ? atoll(v) : 4LL; }();
    if (train_launch_steps < min_steps) return;
    if (overlap_mode() != 2 || !e->cfg.featurize_every_iteration || !e->d_xhat_alt || e->ys_ahead) return;
    (void)hipEventRecord(e->ev_fwd_done, e->stream);
    (void)hipStreamWaitEvent(e->side, e->ev_fwd_done, 0);
    {
        ScopedTimer t(e, ISK_T_FEATURIZE, e->side);
        const long long r0 = e->early_rows;              // already done beside the forward (featurize_early_share)
        featurize_rows(e, e->d_in_ys + r0 * e->Din, e->N * e->K + e->N - r0, e->d_xhat_alt, r0, e->side, e->d_xscale_alt, e->d_xs32_alt,
                       e->N * e->K - r0);
        e->early_rows = 0;
    }
    (void)hipEventRecord(e->ev_ys_ready, e->side);
    e->ys_ahead = true;
}

// Experiment knob ISK_FEAT_EARLY_PCT (default 0 = off): that share of the next iteration's ys rows is featurized on the side
// stream BESIDE THIS ITERATION'S FORWARD (ISK_FEAT_EARLY_WGS workgroups, default 128), the rest beside the training
// kernels as usual: a lighter stream beside the latency-bound training chain at the price of a slower forward.
// Called right before the forward is enqueued (every earlier reader of the other buffers precedes this point).
static void featurize_early_share(isk_engine* e, long long train_launch_steps) {
    static const int pct = [] { const char* v = getenv("ISK_FEAT_EARLY_PCT"); return v ? atoi(v) : 0; }();
    static const int wgs = [] { const char* v = getenv("ISK_FEAT_EARLY_WGS"); return v && atoi(v) > 1 ? atoi(v) : 128; }();
    static const long long min_steps = [] { const char* v = getenv("ISK_SIDE_MIN_STEPS"); return v ? atoll(v) : 4LL; }();
    e->early_rows = 0;
    if (pct <= 0 || train_launch_steps < min_steps) return;
    if (overlap_mode() != 2 || !e->cfg.featurize_every_iteration || !e->d_xhat_alt || e->ys_ahead) return;
    long long r = (e->N * e->K) * (long long)(pct > 100 ? 100 : pct) / 100;
    r -= r % 16;
    if (r <= 0) return;
    (void)hipEventRecord(e->ev_fwd_done, e->stream);
    (void)hipStreamWaitEvent(e->side, e->ev_fwd_done, 0);
    {
        ScopedTimer t(e, ISK_T_FEATURIZE, e->side);
        featurize_rows(e, e->d_in_ys, r, e->d_xhat_alt, 0, e->side, e->d_xscale_alt, nullptr, 0, wgs);
    }
    e->early_rows = r;
}

// The throttled featurizer is the longer branch of [training || featurizer]: start it right behind the forward kernel,
// beside the (tiny, latency-bound) K-mean / target kernels, instead of after them.
static void side_featurize_hook(isk_engine* e) {
    if (e->side_after_forward > 0) {
        featurize_ys_beside_training(e, e->side_after_forward);
        e->side_after_forward = 0;
    }
}

static int set_data_impl(isk_engine* e, const float* xs, const float* ys, int64_t D, int64_t K, int64_t N, hipMemcpyKind kind) {
    if (!e || !xs || !ys) return fail(ISK_EINVAL, "null argument");
    if (D != e->Din) return fail(ISK_EINVAL, "coordinate dimension does not match the featurizer / model input");
    if (K < 1 || N < 1) return fail(ISK_EINVAL, "K and N must be positive");
    if (N > 2000000000ll / 16) return fail(ISK_EUNSUPPORTED, "N too large for 32-bit batch indices");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipStreamSynchronize(e->stream));
    free_data(e);
    e->N = N; e->K = K; e->D = D;
    const long long S = N * K + N;
    const long long Bpad = ((N + 15) / 16) * 16;
    const int H1p = isk_pad16(e->cfg.widths[1]);
    HIPCHK(hipMalloc((void**)&e->d_in_ys, (size_t)(N * K + N) * D * sizeof(float)));
    e->d_in_xs = e->d_in_ys + (size_t)N * K * D;
    const long long Srows = isk_plane_rows(S);           // the unit-blocked plane image (forward_mode 3) holds whole 16-row units
    HIPCHK(hipMalloc((void**)&e->d_xhat, (size_t)Srows * e->Fp * sizeof(float)));
    if (e->cfg.featurize_every_iteration) HIPCHK(hipMalloc((void**)&e->d_xhat_alt, (size_t)Srows * e->Fp * sizeof(float)));
    HIPCHK(hipMalloc((void**)&e->d_xs32, (size_t)N * e->Fp * sizeof(float)));
    if (e->cfg.featurize_every_iteration) HIPCHK(hipMalloc((void**)&e->d_xs32_alt, (size_t)N * e->Fp * sizeof(float)));
    if (e->planes) {
        HIPCHK(hipMalloc((void**)&e->d_xscale, (size_t)S * sizeof(float)));
        if (e->cfg.featurize_every_iteration) HIPCHK(hipMalloc((void**)&e->d_xscale_alt, (size_t)S * sizeof(float)));
    }
    HIPCHK(hipMalloc((void**)&e->d_out, (size_t)S * e->nout * sizeof(float)));
    HIPCHK(hipMalloc((void**)&e->d_ks, (size_t)N * e->nout * sizeof(float)));
    HIPCHK(hipMalloc((void**)&e->d_target, (size_t)N * e->nout * sizeof(float)));
    {
        // K-slice partials of train_fwd1n: KS(B) * pad32(B) rows for any batch size B <= N
        long long zrows = z1_rows(N);
        for (long long RB = 1; (RB - 1) * 32 < N; ++RB) {       // every row-block count a batch of <= N rows can have
            int ks, cpb;
            isk_fwd1n_plan((int)(RB * 32), e->Fp, H1p / 16, &ks, &cpb);
            if (ks > 1) zrows = std::max(zrows, ks * RB * 32); else break;
        }
        HIPCHK(hipMalloc((void**)&e->d_z1, (size_t)zrows * H1p * sizeof(float)));
    }
    HIPCHK(hipMalloc((void**)&e->d_delta1, (size_t)Bpad * H1p * sizeof(float)));
    const TrainShape ts = train_shape(e);
    HIPCHK(hipMalloc((void**)&e->d_part, (size_t)isk_train_part_floats(ts, (int)N) * sizeof(float)));
    HIPCHK(hipMalloc((void**)&e->d_perm, (size_t)N * sizeof(int)));
    HIPCHK(hipHostMalloc((void**)&e->h_perm, (size_t)N * sizeof(int), hipHostMallocDefault));
    if (e->cfg.target_kind == ISK_TARGET_ISA) HIPCHK(hipMalloc((void**)&e->d_isa_work, (size_t)isk_isa_work_doubles(N, e->nout) * sizeof(double)));
    if (e->layered) {
        int mh = 32;
        for (int l = 1; l < e->L; ++l) mh = std::max(mh, isk_pad32(e->cfg.widths[l]));
        HIPCHK(hipMalloc((void**)&e->d_actA, (size_t)S * mh * sizeof(float)));
        HIPCHK(hipMalloc((void**)&e->d_actB, (size_t)S * mh * sizeof(float)));
        HIPCHK(hipMemsetAsync(e->d_actA, 0, (size_t)S * mh * sizeof(float), e->stream));
        HIPCHK(hipMemsetAsync(e->d_actB, 0, (size_t)S * mh * sizeof(float), e->stream));
        for (int l = 0; l < e->L; ++l) {
            const size_t nb = (size_t)Bpad * isk_pad16(e->cfg.widths[l + 1]) * sizeof(float);
            HIPCHK(hipMalloc((void**)&e->d_lz[l], nb));
            HIPCHK(hipMalloc((void**)&e->d_la[l], nb));
            HIPCHK(hipMalloc((void**)&e->d_ld[l], nb));
            HIPCHK(hipMemsetAsync(e->d_lz[l], 0, nb, e->stream));
            HIPCHK(hipMemsetAsync(e->d_la[l], 0, nb, e->stream));
            HIPCHK(hipMemsetAsync(e->d_ld[l], 0, nb, e->stream));
        }
        HIPCHK(hipMalloc((void**)&e->d_xb, (size_t)Bpad * e->Fp * sizeof(float)));
    }
    HIPCHK(hipMemcpyAsync(e->d_in_xs, xs, (size_t)N * D * sizeof(float), kind, e->stream));
    HIPCHK(hipMemcpyAsync(e->d_in_ys, ys, (size_t)N * K * D * sizeof(float), kind, e->stream));
    featurize_resident(e);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipGetLastError());
    return ISK_OK;
}

extern "C" int isk_set_data(isk_engine* e, const float* xs, const float* ys, int64_t D, int64_t K, int64_t N) {
    return set_data_impl(e, xs, ys, D, K, N, hipMemcpyHostToDevice);
}
extern "C" int isk_set_data_dev(isk_engine* e, const float* xs, const float* ys, int64_t D, int64_t K, int64_t N) {
    return set_data_impl(e, xs, ys, D, K, N, hipMemcpyDeviceToDevice);
}
static int need_data(isk_engine* e);

// WeightedSamples(values, weights) (src/data.jl:187-215): weights (1, K, N) column-major == [N][K]
extern "C" int isk_set_koopman_weights(isk_engine* e, const float* w, int64_t K, int64_t N) {
    CHK(need_data(e));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (!w) { free_dev(e->d_kw); e->d_kw = nullptr; return ISK_OK; }
    if (K != e->K || N != e->N) return fail(ISK_EINVAL, "weights must be (1, K, N) for the resident data");
    if (!e->d_kw) HIPCHK(hipMalloc((void**)&e->d_kw, (size_t)N * K * sizeof(float)));
    HIPCHK(hipMemcpyAsync(e->d_kw, w, (size_t)N * K * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->have_ks = e->have_target = false;
    return ISK_OK;
}

extern "C" int isk_data_shape(isk_engine* e, int64_t* D, int64_t* K, int64_t* N) {
    if (!e) return fail(ISK_EINVAL, "null engine");
    if (D) *D = e->D; if (K) *K = e->K; if (N) *N = e->N;
    return ISK_OK;
}

// ---------------------------------------------------------------------------------------
// featurizer / forward on caller-provided columns (not the resident data)
// ---------------------------------------------------------------------------------------
static int launch_forward(isk_engine* e, const MlpFwdArgs& f);

struct TmpBuf {
    void* p = nullptr;
    ~TmpBuf() { if (p) (void)hipFree(p); }
};

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
    a.planes = e->planes; a.xscale = (float*)xsc.p;
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

// ---------------------------------------------------------------------------------------
// resident forward, koopman, target
// ---------------------------------------------------------------------------------------
// bf16 planes per f32 operand of the layer-1 / dense-block forward GEMM: 0 = f32 MFMA, 3 = split-bf16, 1 = plain bf16
static int bf16_planes(const isk_engine* e) { return e->cfg.forward_mode == 1 ? 3 : (e->cfg.forward_mode == 2 ? 1 : 0); }

// fused networks: refresh the bf16 weight planes now (launch_forward would do it lazily, i.e. behind whatever the
// stream is about to wait for)
static void fold_ahead(isk_engine* e) {
    ensure_folded(e);
    const int planes = bf16_planes(e);
    if (e->layered || !(planes || e->planes) || !e->w16_dirty) return;
    if (e->planes) isk_launch_fold_f16(e->d_Wt[0], e->d_W16, e->Fp, isk_pad16(e->cfg.widths[1]) / 16, e->d_wmax + e->wmax_last, e->d_wscale, e->stream);
    else isk_launch_fold_bf16(e->d_Wt[0], e->d_W16, e->Fp, isk_pad16(e->cfg.widths[1]) / 16, planes, e->stream);
    e->w16_dirty = false;
}

// layer-by-layer forward: every layer is a sequence of dense-block launches of the same GEMM kernels; hidden
// activations [S][pad32(H)] ping-pong between actA / actB (zero-initialised: the padding columns stay 0)
static int forward_layered(isk_engine* e, const float* X, long long S, float* out, float* actA, float* actB) {
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
        static const int grouped = [] { const char* v = getenv("ISK_DENSE_GROUPS"); return v ? atoi(v) : 1; }();
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

// the forward over feature rows: f32 MFMA or split-bf16 (cfg.forward_mode)
static int launch_forward(isk_engine* e, const MlpFwdArgs& f) {
    ensure_folded(e);
    if (e->layered) {
        float *a = e->d_actA, *b = e->d_actB;
        TmpBuf ta, tb;
        const bool resident = (f.X >= e->d_xhat && e->d_xhat && f.X < e->d_xhat + (long long)(e->N * e->K + e->N) * e->Fp) ||
                              (e->d_xhat_alt && f.X >= e->d_xhat_alt && f.X < e->d_xhat_alt + (long long)(e->N * e->K + e->N) * e->Fp);
        if (!resident || !a) {
            int mh = 32;
            for (int l = 1; l < e->L; ++l) mh = std::max(mh, isk_pad32(e->cfg.widths[l]));
            if (hipMalloc(&ta.p, (size_t)f.S * mh * sizeof(float)) != hipSuccess || hipMalloc(&tb.p, (size_t)f.S * mh * sizeof(float)) != hipSuccess)
                return ISK_ENOMEM;
            (void)hipMemsetAsync(ta.p, 0, (size_t)f.S * mh * sizeof(float), e->stream);
            (void)hipMemsetAsync(tb.p, 0, (size_t)f.S * mh * sizeof(float), e->stream);
            a = (float*)ta.p; b = (float*)tb.p;
            const int r = forward_layered(e, f.X, f.S, f.out, a, b);
            (void)hipStreamSynchronize(e->stream);     // the temporaries die with this scope
            return r;
        }
        return forward_layered(e, f.X, f.S, f.out, a, b);
    }
    if (e->planes) {
        // the rows are f16 planes: no other kernel can read them
        if (!f.xscale) return ISK_EUNSUPPORTED;
        fold_ahead(e);
        return isk_launch_mlp_fwd_f16x2(f, e->d_W16, e->stream);
    }
    if (const int planes = bf16_planes(e)) {
        if (e->w16_dirty) {
            isk_launch_fold_bf16(e->d_Wt[0], e->d_W16, e->Fp, f.nt1, planes, e->stream);
            e->w16_dirty = false;
        }
        if (isk_launch_mlp_fwd_bf16x3(f, e->d_W16, planes, e->stream) == 0) return 0;
        // shapes whose LDS footprint does not fit the split-bf16 kernel run on the f32 kernel
    }
    return isk_launch_mlp_fwd(f, e->stream);
}

static int need_data(isk_engine* e) {
    if (!e) return fail(ISK_EINVAL, "null engine");
    if (e->N == 0) return fail(ISK_ESTATE, "no data: call isk_set_data first");
    HIPCHK(hipSetDevice(e->cfg.device));
    return ISK_OK;
}

// which: 1 = ys rows, 2 = xs rows, 3 = both (single launch over the contiguous slab)
static int forward_resident(isk_engine* e, int which, bool ahead = false) {
    const long long nys = e->N * e->K;
    const float* X = e->d_xhat;
    float* out = e->d_out;
    long long S = nys + e->N;
    int tid = ISK_T_FWD_YS;
    if (which == 1) S = nys;
    if (which == 2) { if (!e->planes) X += nys * e->Fp; out += nys * e->nout; S = e->N; tid = ISK_T_FWD_XS; }
    if (ahead) (void)hipEventRecord(e->ev_fwd_done, e->stream);
    {
        ScopedTimer t(e, tid);
        MlpFwdArgs f = fwd_args(e, X, S, out);
        if (e->planes) { f.xscale = e->d_xscale + (which == 2 ? nys : 0); f.x_row0 = which == 2 ? nys : 0; }   // unit-blocked rows: base + row index
        // diagnostic only (ISK_DEBUG_STAMPS=1 with a -DISK_STAMP build of the bf16x3 kernel): phase cycle sums
        static const bool stamps = getenv("ISK_DEBUG_STAMPS") != nullptr;
        static unsigned long long* dbg_st = nullptr;
        if (stamps) {
            if (!dbg_st) (void)hipMalloc((void**)&dbg_st, (size_t)4096 * 12 * 8);
            (void)hipMemsetAsync(dbg_st, 0, (size_t)4096 * 12 * 8, e->stream);
            f.dbg = dbg_st;
        }
        // diagnostic only (ISK_DEBUG_ROWMOD=m): every row reads row (r mod m), i.e. a cache-resident sample stream
        static const long long rowmod = [] { const char* v = getenv("ISK_DEBUG_ROWMOD"); return v ? atoll(v) : 0LL; }();
        static int* dbg_rows = nullptr;
        if (rowmod > 0) {
            if (!dbg_rows) {
                std::vector<int> h(S);
                for (long long i = 0; i < S; ++i) h[i] = (int)(i % rowmod);
                (void)hipMalloc((void**)&dbg_rows, (size_t)S * sizeof(int));
                (void)hipMemcpy(dbg_rows, h.data(), (size_t)S * sizeof(int), hipMemcpyHostToDevice);
            }
            f.rows = dbg_rows;
        }
        if (launch_forward(e, f) != 0) return fail(ISK_EUNSUPPORTED, "model shape not covered by the forward kernel");
        if (stamps) {
            (void)hipStreamSynchronize(e->stream);
            std::vector<unsigned long long> h((size_t)4096 * 12);
            (void)hipMemcpy(h.data(), dbg_st, h.size() * 8, hipMemcpyDeviceToHost);
            double sum[10] = {0}; int nw = 0;
            for (int w = 0; w < 4096; ++w) {
                if (!h[(size_t)w * 12 + 8]) continue;
                ++nw;
                for (int k = 0; k < 10; ++k) sum[k] += (double)h[(size_t)w * 12 + k];
            }
            if (nw) {
                fprintf(stderr, "[stamps] waves %d  total cyc %.0f  realtime(100MHz ticks) %.0f -> clock %.2f GHz\n", nw, sum[8] / nw, sum[9] / nw,
                        sum[8] / sum[9] * 0.1);
                const char* nm[8] = {"pass prologue", "stage: up to barrier", "barrier", "stage: after barrier", "end-of-pass barrier", "epilogue", "-", "-"};
                for (int k = 0; k < 8; ++k) fprintf(stderr, "[stamps]   %-20s %6.1f %%\n", nm[k], 100.0 * sum[k] / sum[8]);
            }
        }
    }
    if (ahead) featurize_ys_ahead(e);
    return ISK_OK;
}

static void launch_kmean(isk_engine* e) {
    KmeanArgs k;
    k.out_ys = e->d_out; k.ks = e->d_ks; k.w = e->d_kw; k.N = e->N; k.K = (int)e->K; k.nout = e->nout;
    k.zero2 = e->d_loss_acc;             // the coming epoch's loss accumulator: zeroed here instead of by a memset node of its own
    isk_launch_kmean(k, e->stream);
    e->loss_acc_zeroed = true;
    e->have_ks = true;
}

extern "C" int isk_chis(isk_engine* e, float* out) {
    CHK(need_data(e));
    if (!out) return fail(ISK_EINVAL, "null argument");
    CHK(forward_resident(e, 2));
    HIPCHK(hipMemcpyAsync(out, e->d_out + e->N * e->K * e->nout, (size_t)e->N * e->nout * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipGetLastError());
    return ISK_OK;
}

extern "C" int isk_koopman(isk_engine* e, float* out) {
    CHK(need_data(e));
    if (!out) return fail(ISK_EINVAL, "null argument");
    CHK(forward_resident(e, 1));
    launch_kmean(e);
    HIPCHK(hipMemcpyAsync(out, e->d_ks, (size_t)e->N * e->nout * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipGetLastError());
    return ISK_OK;
}

// ---- host Float64 ISA (src/isotarget.jl:81-127; PCCAPlus.indexmap restated) -----------
static std::vector<int> indexmap_host(std::vector<double> X /* N x d row-major */, long long N, int d) {
    std::vector<int> ind(d);
    std::vector<double> rn(N);
    for (int j = 0; j < d; ++j) {
        long long best = 0;
        double bv = -1.0;
        for (long long n = 0; n < N; ++n) {
            double s = 0;
            for (int k = 0; k < d; ++k) s += X[n * d + k] * X[n * d + k];
            rn[n] = std::sqrt(s);
            if (rn[n] > bv) { bv = rn[n]; best = n; }      // first maximum, like Julia's argmax
        }
        ind[j] = (int)best;
        if (j == 0) {
            std::vector<double> x0(X.begin() + best * d, X.begin() + best * d + d);
            for (long long n = 0; n < N; ++n)
                for (int k = 0; k < d; ++k) X[n * d + k] -= x0[k];
        } else {
            const double inv = 1.0 / rn[best];
            for (auto& v : X) v *= inv;
            std::vector<double> vt(X.begin() + best * d, X.begin() + best * d + d);
            for (long long n = 0; n < N; ++n) {
                double dot = 0;
                for (int k = 0; k < d; ++k) dot += X[n * d + k] * vt[k];
                for (int k = 0; k < d; ++k) X[n * d + k] -= dot * vt[k];
            }
        }
    }
    return ind;
}

// Gauss-Jordan inverse with partial pivoting; false if singular / non-finite
static bool invert_host(std::vector<double>& A, int d) {
    std::vector<double> M(d * 2 * d, 0.0);
    for (int i = 0; i < d; ++i) {
        for (int j = 0; j < d; ++j) M[i * 2 * d + j] = A[i * d + j];
        M[i * 2 * d + d + i] = 1.0;
    }
    // multiplier against the unscaled pivot row (as LU does): exactly dependent rows cancel to exact
    // zeros and are reported singular, like Julia's inv()
    for (int c = 0; c < d; ++c) {
        int p = c;
        for (int r = c + 1; r < d; ++r)
            if (std::fabs(M[r * 2 * d + c]) > std::fabs(M[p * 2 * d + c])) p = r;
        const double pv = M[p * 2 * d + c];
        if (!(std::fabs(pv) > 0.0) || !std::isfinite(pv)) return false;
        if (p != c)
            for (int j = 0; j < 2 * d; ++j) std::swap(M[p * 2 * d + j], M[c * 2 * d + j]);
        for (int r = 0; r < d; ++r) {
            if (r == c) continue;
            const double f = M[r * 2 * d + c] / pv;
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * d; ++j) M[r * 2 * d + j] -= f * M[c * 2 * d + j];
            M[r * 2 * d + c] = 0.0;
        }
    }
    for (int c = 0; c < d; ++c) {
        const double inv = 1.0 / M[c * 2 * d + c];
        for (int j = 0; j < 2 * d; ++j) M[c * 2 * d + j] *= inv;
    }
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            A[i * d + j] = M[i * 2 * d + d + j];
            if (!std::isfinite(A[i * d + j])) return false;
        }
    return true;
}

// symmetric C -> C^(-1/2) by cyclic Jacobi (whitening option, src/isotarget.jl:85-89)
static bool inv_sqrt_sym(std::vector<double>& C, int d) {
    std::vector<double> V(d * d, 0.0), A = C;
    for (int i = 0; i < d; ++i) V[i * d + i] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0;
        for (int i = 0; i < d; ++i)
            for (int j = i + 1; j < d; ++j) off += A[i * d + j] * A[i * d + j];
        if (off < 1e-300) break;
        for (int p = 0; p < d; ++p)
            for (int q = p + 1; q < d; ++q) {
                if (std::fabs(A[p * d + q]) < 1e-300) continue;
                const double theta = (A[q * d + q] - A[p * d + p]) / (2 * A[p * d + q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                const double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < d; ++k) {
                    const double akp = A[k * d + p], akq = A[k * d + q];
                    A[k * d + p] = c * akp - s * akq; A[k * d + q] = s * akp + c * akq;
                }
                for (int k = 0; k < d; ++k) {
                    const double apk = A[p * d + k], aqk = A[q * d + k];
                    A[p * d + k] = c * apk - s * aqk; A[q * d + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < d; ++k) {
                    const double vkp = V[k * d + p], vkq = V[k * d + q];
                    V[k * d + p] = c * vkp - s * vkq; V[k * d + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < d; ++i)
        if (!(A[i * d + i] > 0)) return false;
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            double s = 0;
            for (int k = 0; k < d; ++k) s += V[i * d + k] * V[j * d + k] / std::sqrt(A[k * d + k]);
            C[i * d + j] = s;
        }
    return true;
}

// ks, chi: [N][d] (fp32, as downloaded).  Produces At (A' row-major: target = At * ks[n]) and the row permutation.
static int isa_solve_host(const isk_engine* e, const std::vector<float>& ks, const std::vector<float>& chi, long long N, int d,
                          std::vector<double>& At, std::vector<int>& rowperm, std::vector<double>* target_out) {
    for (float v : ks)
        if (!std::isfinite(v)) return fail(ISK_ESINGULAR, isk_status_string(ISK_ESINGULAR));
    std::vector<double> X(ks.begin(), ks.end());           // X = Float64.(ks')  (N x d)
    std::vector<int> ind;
    if (e->cfg.isa_whitening) {
        std::vector<double> C(d * d, 0.0);
        for (long long n = 0; n < N; ++n)
            for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) C[i * d + j] += X[n * d + i] * X[n * d + j];
        for (auto& v : C) v /= (double)N;
        if (!inv_sqrt_sym(C, d)) return fail(ISK_ESINGULAR, isk_status_string(ISK_ESINGULAR));
        std::vector<double> XW((size_t)N * d);
        for (long long n = 0; n < N; ++n)
            for (int j = 0; j < d; ++j) {
                double s = 0;
                for (int k = 0; k < d; ++k) s += X[n * d + k] * C[k * d + j];
                XW[n * d + j] = s;
            }
        ind = indexmap_host(XW, N, d);
    } else {
        ind = indexmap_host(X, N, d);
    }
    std::vector<double> A(d * d);
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) A[i * d + j] = X[(long long)ind[i] * d + j];
    if (!invert_host(A, d)) return fail(ISK_ESINGULAR, isk_status_string(ISK_ESINGULAR));
    // target = A' * ks  ->  target[n][a] = sum_b A[b][a] ks[n][b]
    At.assign(d * d, 0.0);
    for (int a = 0; a < d; ++a)
        for (int b = 0; b < d; ++b) At[a * d + b] = A[b * d + a];
    rowperm.resize(d);
    std::iota(rowperm.begin(), rowperm.end(), 0);
    std::vector<double> tgt((size_t)N * d);
    for (long long n = 0; n < N; ++n)
        for (int a = 0; a < d; ++a) {
            double s = 0;
            for (int b = 0; b < d; ++b) s += At[a * d + b] * X[n * d + b];
            tgt[n * d + a] = s;
        }
    if (e->cfg.isa_permute) {
        // fixperm (src/isotarget.jl:120-127): cost(p) = sum_i sum_n |new[p_i, n] - old[i, n]|
        std::vector<double> Cst(d * d, 0.0);
        for (long long n = 0; n < N; ++n)
            for (int a = 0; a < d; ++a)
                for (int i = 0; i < d; ++i) Cst[a * d + i] += std::fabs(tgt[n * d + a] - (double)chi[n * d + i]);
        std::vector<int> p(d), best(d);
        std::iota(p.begin(), p.end(), 0);
        double bc = INFINITY;
        bool first = true;
        do {
            double c = 0;
            for (int i = 0; i < d; ++i) c += Cst[p[i] * d + i];
            if (first || c < bc) { bc = c; best = p; first = false; }
        } while (std::next_permutation(p.begin(), p.end()));
        rowperm = best;
    }
    if (target_out) {
        target_out->resize((size_t)N * d);
        for (long long n = 0; n < N; ++n)
            for (int i = 0; i < d; ++i) (*target_out)[n * d + i] = tgt[n * d + rowperm[i]];
    }
    return ISK_OK;
}

// loss weights 1 ./ std(target, dims=2) (corrected), src/iso.jl:183
static void loss_weights_host(const std::vector<float>& tgt, long long N, int d, std::vector<float>& w) {
    w.assign(d, 1.0f);
    if (d <= 1) return;
    for (int i = 0; i < d; ++i) {
        double mean = 0;
        for (long long n = 0; n < N; ++n) mean += tgt[n * d + i];
        mean /= (double)N;
        double ss = 0;
        for (long long n = 0; n < N; ++n) { const double t = tgt[n * d + i] - mean; ss += t * t; }
        w[i] = (float)(1.0 / std::sqrt(ss / (double)(N - 1)));
    }
}

static int check_status(isk_engine* e) {
    int st = 0;
    HIPCHK(hipMemcpyAsync(&st, e->d_status, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (st) HIPCHK(hipMemsetAsync(e->d_status, 0, sizeof(int), e->stream));
    if (st & 1) return fail(ISK_ECONSTANT_CHI, isk_status_string(ISK_ECONSTANT_CHI));
    if (st & 2) return fail(ISK_ESINGULAR, isk_status_string(ISK_ESINGULAR));
    return ISK_OK;
}

// computes the target on the device-resident data; leaves target + loss weights on the device
static int compute_target(isk_engine* e, bool check, bool ahead = false) {
    const int d = e->nout;
    if (e->cfg.target_kind == ISK_TARGET_HOST)
        return fail(ISK_EINVAL, "host-side target transform: compute it from isk_chis / isk_koopman and pass it to isk_train_batch");
    if (e->cfg.target_kind == ISK_TARGET_SHIFTSCALE) {
        CHK(forward_resident(e, 1, ahead));
        side_featurize_hook(e);
        ScopedTimer t(e, ISK_T_TARGET);
        launch_kmean(e);
        isk_launch_minmax(e->d_ks, e->N, e->d_mm, e->stream);
        isk_launch_shiftscale(e->d_ks, e->d_mm, e->d_target, e->N, e->d_status, e->stream);
        e->have_target = true;
        if (check) CHK(check_status(e));
        return ISK_OK;
    }
    // ISA: chi = model(xs), ks = E[model(ys)] in one launch over the contiguous slab, then the
    // Float64 simplex solve, fixperm and the loss weights in a single-workgroup kernel: the
    // iteration never leaves the device (the reference pulls chi and ks to the host here,
    // src/isotarget.jl:101,103)
    CHK(forward_resident(e, 3, ahead));
    side_featurize_hook(e);
    ScopedTimer t(e, ISK_T_TARGET);
    launch_kmean(e);
    IsaArgs ia;
    ia.ks = e->d_ks; ia.chi = e->d_out + e->N * e->K * d; ia.work = e->d_isa_work; ia.target = e->d_target;
    ia.lossw = e->d_lossw; ia.status = e->d_status; ia.N = e->N; ia.d = d;
    ia.permute = e->cfg.isa_permute; ia.whitening = e->cfg.isa_whitening;
    ia.out_first = 0; ia.out_count = e->N;
    isk_launch_isa(ia, e->stream);
    e->have_target = true;
    if (check) CHK(check_status(e));
    return ISK_OK;
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
    a.planes = e->planes; a.xscale = (float*)xsc.p;
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
    if (e->Fp > 2048 || D > 2048) return fail(ISK_EUNSUPPORTED, "pullback covers <= 2048 features / coordinates per sample");
    const int H1p = isk_pad16(e->cfg.widths[1]);
    const TrainShape ts = train_shape(e);
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

extern "C" int isk_target(isk_engine* e, float* out) {
    CHK(need_data(e));
    CHK(compute_target(e, true));
    if (out) {
        HIPCHK(hipMemcpyAsync(out, e->d_target, (size_t)e->N * e->nout * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    HIPCHK(hipGetLastError());
    return ISK_OK;
}

// ---------------------------------------------------------------------------------------
// training
// ---------------------------------------------------------------------------------------
// nslab > 1: the layer-1 gradient GEMM splits the batch rows over `nslab` workgroups per feature tile
// (twice the CUs, half the MFMA chain each); the optimizer kernel sums the slabs.  The data-parallel
// path keeps nslab = 1 because the flat all-reduce buffer is slab 0.
static int pick_slabs(int B) {
    const int nchunks = (B + 15) / 16;
    int s = (nchunks + 47) / 48;                  // about three 16-row chunks per wave
    return s < 1 ? 1 : (s > ISK_MAX_SLABS ? ISK_MAX_SLABS : s);
}

// ---------------------------------------------------------------------------------------
// rocBLAS for the PLAIN GEMMs of the layered (wide-network) training path, loaded lazily.  The fused kernels
// cover the networks whose layers fit a workgroup; 512-wide layers at a 1000-row batch are ordinary
// [1000 x 2016] x [2016 x 512] products, where a tiled library GEMM moves 4-8x fewer bytes through L2 than
// the one-tile-per-workgroup row GEMM used before (C5 shard: 0.47 -> ~0.2 ms per optimizer step).
// ISK_LAYERED_BLAS=0 keeps the in-house kernels (also the fallback when librocblas.so cannot be loaded).
// ---------------------------------------------------------------------------------------
namespace {
struct BlasApi {
    void* lib = nullptr;
    rocblas_status (*create)(rocblas_handle*) = nullptr;
    rocblas_status (*destroy)(rocblas_handle) = nullptr;
    rocblas_status (*set_stream)(rocblas_handle, hipStream_t) = nullptr;
    rocblas_status (*sgemm)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int, rocblas_int, const float*,
                            const float*, rocblas_int, const float*, rocblas_int, const float*, float*, rocblas_int) = nullptr;
    bool failed = false;
};
BlasApi g_blas;

bool blas_load() {
    if (g_blas.lib) return true;
    if (g_blas.failed) return false;
    void* h = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librocblas.so", RTLD_NOW | RTLD_GLOBAL);
    if (h) {
        g_blas.create = (decltype(g_blas.create))dlsym(h, "rocblas_create_handle");
        g_blas.destroy = (decltype(g_blas.destroy))dlsym(h, "rocblas_destroy_handle");
        g_blas.set_stream = (decltype(g_blas.set_stream))dlsym(h, "rocblas_set_stream");
        g_blas.sgemm = (decltype(g_blas.sgemm))dlsym(h, "rocblas_sgemm");
    }
    if (!h || !g_blas.create || !g_blas.destroy || !g_blas.set_stream || !g_blas.sgemm) { g_blas.failed = true; return false; }
    g_blas.lib = h;
    return true;
}
}  // namespace

static void blas_release(isk_engine* e) {
    if (e->blas && g_blas.destroy) (void)g_blas.destroy(e->blas);
    e->blas = nullptr;
}

static bool blas_ready(isk_engine* e) {
    static const int on = [] { const char* v = getenv("ISK_LAYERED_BLAS"); return v ? atoi(v) : 1; }();
    if (!on || !e->layered || !e->d_xb) return false;
    if (!blas_load()) return false;
    if (!e->blas && g_blas.create(&e->blas) != rocblas_status_success) { e->blas = nullptr; g_blas.failed = true; return false; }
    return g_blas.set_stream(e->blas, e->stream) == rocblas_status_success;
}

// one optimizer step's gradient on the layered path with library GEMMs.  All matrices are viewed column-major:
// a row-major [rows][ld] activation buffer is the (ld x rows) matrix X', Flux's W (out, in) is the (N x K) matrix W.
//   forward   Z' (N x B) = W (N x K) * X' (K x B)
//   dW        G  (N x K) = D' (N x B) * X'^T          -> written in Flux layout directly
//   dX        R' (K x B) = W^T (K x N) * D' (N x B)   -> no transposed weight image needed
static bool launch_grads_layered_blas(isk_engine* e, const int* d_idx, int B, float inv_div) {
    if (!blas_ready(e)) return false;
    const int L = e->L;
    const long long Bpad = ((B + 15) / 16) * 16;
    const float* xs = e->d_xs32;
    hipStream_t st = e->stream;
    const float one = 1.f, zero = 0.f;
    const float* X0 = xs;
    if (d_idx) { isk_launch_gather_rows(xs, e->Fp, d_idx, B, e->d_xb, st); X0 = e->d_xb; }
    const float* W0 = e->d_theta + e->off_W[0];
    if (e->cfg.layernorm) {
        isk_launch_scale_cols(e->d_theta + e->off_W[0], e->d_theta + e->off_lns, e->d_W0g, e->cfg.widths[1], e->F, st);
        W0 = e->d_W0g;
    }
    bool ok = true;
    for (int l = 0; l < L && ok; ++l) {
        const int K = e->cfg.widths[l], N = e->cfg.widths[l + 1], Np = isk_pad16(N);
        const float* X = (l == 0) ? X0 : e->d_la[l - 1];
        const int ldx = (l == 0) ? e->Fp : isk_pad16(K);
        const float* W = (l == 0) ? W0 : e->d_theta + e->off_W[l];
        ok = g_blas.sgemm(e->blas, rocblas_operation_none, rocblas_operation_none, N, B, K, &one, W, N, X, ldx, &zero, e->d_lz[l], Np) ==
             rocblas_status_success;
        // layer 0 with LayerNorm: the folded bias image (b1 + W1*beta); otherwise the bias straight from theta
        const float* bias = (l == 0 && e->cfg.layernorm) ? e->d_bias[0] : e->d_theta + e->off_b[l];
        isk_launch_bias_act(e->d_lz[l], bias, l < L - 1 ? e->d_la[l] : nullptr, B, Np, N, act_of_layer(e, l), st);
    }
    if (!ok) return false;
    const int nop = isk_pad16(e->nout);
    isk_launch_loss_delta(e->d_lz[L - 1], nop, e->d_target, e->d_lossw, d_idx, B, e->nout, inv_div, e->cfg.lastactivation,
                          e->d_ld[L - 1], nop, e->d_grad + e->nparams, st);
    for (int l = L - 1; l >= 0 && ok; --l) {
        const int K = e->cfg.widths[l], N = e->cfg.widths[l + 1], Np = isk_pad16(N);
        const float* X = (l == 0) ? X0 : e->d_la[l - 1];
        const int ldx = (l == 0) ? e->Fp : isk_pad16(K);
        const bool ln0 = (l == 0 && e->cfg.layernorm);
        float* gW = ln0 ? e->d_G0 : e->d_grad + e->off_W[l];
        ok = g_blas.sgemm(e->blas, rocblas_operation_none, rocblas_operation_transpose, N, K, B, &one, e->d_ld[l], Np, X, ldx, &zero, gW, N) ==
             rocblas_status_success;
        isk_launch_colsum(e->d_ld[l], Np, B, N, e->d_grad + e->off_b[l], st);
        if (ln0)
            isk_launch_ln_unfold(e->d_G0, e->d_grad + e->off_b[0], e->d_theta, e->off_W[0], e->off_lns, e->off_lnb, e->F, N, e->d_grad, st);
        if (l > 0 && ok) {
            // delta_{l-1} = (delta_l W_l) .* act'(z_{l-1});  a_{l-1} is dead after dW_l and receives the raw product
            ok = g_blas.sgemm(e->blas, rocblas_operation_transpose, rocblas_operation_none, K, B, N, &one, e->d_theta + e->off_W[l], N,
                              e->d_ld[l], Np, &zero, e->d_la[l - 1], isk_pad16(K)) == rocblas_status_success;
            isk_launch_dact_mul(e->d_la[l - 1], e->d_lz[l - 1], e->d_ld[l - 1], Bpad, isk_pad16(K), K, act_of_layer(e, l - 1), st);
        }
    }
    e->last_nslab = 1;
    return ok;
}

// layered path: one optimizer step's gradient, layer by layer (see kernels_layered.hip)
static void launch_grads_layered(isk_engine* e, const int* d_idx, int B, float inv_div) {
    if (launch_grads_layered_blas(e, d_idx, B, inv_div)) return;
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
                          e->d_ld[L - 1], nop, e->d_grad + e->nparams, st);
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

// bring the device copy of [beta1^t, beta2^t] up to date (after create / set_opt_state / reset, or steps taken in host mode);
// synchronous and rare.  Never called while the stream is being captured.
static bool prepare_dev_bt(isk_engine* e) {
    if (e->bt_dev_valid) return true;
    const double h[2] = {e->bt1, e->bt2};
    if (hipMemcpyAsync(e->d_bt, h, sizeof(h), hipMemcpyHostToDevice, e->stream) != hipSuccess) return false;
    if (hipStreamSynchronize(e->stream) != hipSuccess) return false;
    e->bt_dev_valid = true;
    return true;
}

// one optimizer step's gradient into d_grad (slab 0 .. nslab-1).  step: an optimizer step follows (launch_opt), so
// train_tail hands the Adam bias corrections of this step over on the device
static void launch_grads(isk_engine* e, const int* d_idx, int B, float inv_div, int nslab = 1, bool step = true, bool capturing = false) {
    if (e->layered) { launch_grads_layered(e, d_idx, B, inv_div); return; }
    const TrainShape s = train_shape(e);
    TrainBuffers b = train_buffers(e);
    int cpb;
    isk_fwd1n_plan(B, e->Fp, isk_pad16(e->cfg.widths[1]) / 16, &b.KS, &cpb);
    b.zstride = z1_rows(B) * isk_pad16(e->cfg.widths[1]);
    if (step && (capturing ? e->bt_dev_valid : prepare_dev_bt(e))) b.bt = e->d_bt;
    isk_launch_train_fwd1n(s, b, d_idx, B, e->stream);
    isk_launch_train_tail(s, b, d_idx, B, inv_div, e->stream);
    // while the throttled featurizer shares the CUs, use the 8-wave variant (80 KB LDS instead of 158 KB) so both fit
    isk_launch_train_grad1n(s, b, d_idx, B, nslab | (e->ys_ahead ? 0x100 : 0), e->stream);
    e->last_nslab = nslab;
}

// fold: refresh the weight images the forward / training kernels read.  The rocBLAS variant of the layered training
// path reads theta itself (only layer 0's folded bias comes from an image, and only with LayerNorm), so inside an
// epoch it skips the refresh and folds once at the end.
// publish: this is the epoch's LAST step -- it writes the result into mapped pinned host memory (a kernel ends only when its
// stores are acknowledged: publishing from every step cost ~2 us per step, 0.2 ms per iteration at minibatch = 100)
static void launch_opt(isk_engine* e, float gscale, bool fold = true, bool capturing = false, bool publish = true) {
    OptArgs a;
    a.theta = e->d_theta; a.grad = e->d_grad; a.m = e->d_m; a.v = e->d_v; a.n = e->nparams;
    a.kind = e->cfg.opt_kind; a.lr = e->cfg.lr; a.reg = e->cfg.reg; a.beta1 = e->cfg.beta1; a.beta2 = e->cfg.beta2;
    a.eps = e->cfg.opt_eps; a.rho = e->cfg.rho;
    a.bc1 = (float)(1.0 - e->bt1); a.bc2 = (float)(1.0 - e->bt2);
    // fused path: the bias corrections of this step were handed over on the device by train_tail (launch_grads)
    const bool dev = !e->layered && e->bt_dev_valid;
    (void)capturing;
    a.bt = dev ? e->d_bt + 2 : nullptr;
    a.gscale = gscale; a.loss_acc = e->d_loss_acc;
    a.host_pub = publish ? e->d_hres : nullptr;
    a.nslab = e->last_nslab;
    a.slab_prefix = e->off_b[0] + e->cfg.widths[1];          // [LN | W1 | b1]: the blocks train_grad1n writes per row slab
    isk_launch_opt_step(a, e->stream);
    e->loss_acc_zeroed = false;
    e->bt1 *= e->cfg.beta1; e->bt2 *= e->cfg.beta2;
    if (!dev) e->bt_dev_valid = false;           // this step advanced only the host copy
    if (e->native) { e->wt_dirty = true; e->w16_dirty = true; }      // the training steps read theta; forwards re-fold on demand
    else if (fold) launch_fold(e);
}

static int upload_target(isk_engine* e, const float* target) {
    HIPCHK(hipMemcpyAsync(e->d_target, target, (size_t)e->N * e->nout * sizeof(float), hipMemcpyHostToDevice, e->stream));
    std::vector<float> tf(target, target + (size_t)e->N * e->nout), w;
    loss_weights_host(tf, e->N, e->nout, w);
    HIPCHK(hipMemcpyAsync(e->d_lossw, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->have_target = true;
    return ISK_OK;
}

// one epoch on the device-resident target; perm on host or nullptr (engine shuffle)
// isk_run uploads the first epoch's shuffle order on the side stream while the forward runs: a host->device copy
// in front of the first training kernel costs ~20 us of idle GPU (SDMA latency) on the main stream
static int stage_perm_ahead(isk_engine* e, const int32_t* perm) {
    const long long N = e->N;
    if (perm) {
        for (long long i = 0; i < N; ++i)
            if (perm[i] < 0 || perm[i] >= N) return fail(ISK_EINVAL, "permutation entry out of range");
        std::memcpy(e->h_perm, perm, (size_t)N * sizeof(int));
    } else {
        std::vector<int32_t> p(N);
        shuffle_perm(e, p);
        std::memcpy(e->h_perm, p.data(), (size_t)N * sizeof(int));
    }
    HIPCHK(hipMemcpyAsync(e->d_perm, e->h_perm, (size_t)N * sizeof(int), hipMemcpyHostToDevice, e->copy));
    HIPCHK(hipEventRecord(e->ev_perm, e->copy));
    return ISK_OK;
}

// The epoch's nb x 5 dependent launches as ONE hipGraph launch.  A dependent kernel launched from the host costs ~3.0 us
// on this stack, the same chain replayed from a graph ~1.9 us (measured with a 50-kernel chain of empty kernels); the
// training chain is nothing but such launches (50 per iteration at C3, 500 at the reference's default minibatch of 100).
// Every launch argument is constant from one epoch to the next: the batch slices are fixed offsets into d_perm (its
// CONTENT is what the shuffle changes), and the Adam bias corrections are read from d_bt on the device.  The graph is
// captured from the ordinary launch functions the first time and re-captured when the batch plan, the train_grad1
// variant or the buffers change.  ISK_TRAIN_GRAPH=0 (or any capture error) keeps the per-kernel launches.
static bool train_epoch_graph(isk_engine* e, long long bs, long long nb) {
    static const int on = [] { const char* v = getenv("ISK_TRAIN_GRAPH"); return v ? atoi(v) : 1; }();
    if (!on || e->tg_disabled || e->layered || nb < 2) return false;
    const int nslab = pick_slabs((int)bs);
    const int flags = nslab | (e->ys_ahead ? 0x100 : 0);
    if (!prepare_dev_bt(e)) return false;
    const float* xs_rows = e->d_xs32;                    // part of the key: the buffer pair swaps every iteration
    isk_engine::TrainGraph* tg = nullptr;
    for (auto& g : e->tg)
        if (g.exec && g.xs == xs_rows) tg = &g;
    if (!tg) {                                           // a free slot, else the one that does not belong to the other live buffer
        tg = !e->tg[0].exec ? &e->tg[0] : (!e->tg[1].exec ? &e->tg[1] : (e->tg[0].xs == e->d_xs32_alt ? &e->tg[1] : &e->tg[0]));
    }
    if (!tg->exec || tg->bs != bs || tg->nb != nb || tg->flags != flags || tg->xs != xs_rows) {
        if (tg->exec) { (void)hipGraphExecDestroy(tg->exec); tg->exec = nullptr; }
        const double s1 = e->bt1, s2 = e->bt2;
        if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            e->tg_disabled = true;
            return false;
        }
        for (long long b = 0; b < nb; ++b) {
            launch_grads(e, e->d_perm + b * bs, (int)bs, 1.0f / (float)bs, nslab, true, true);
            launch_opt(e, 1.0f, true, true, b == nb - 1);
        }
        hipGraph_t g = nullptr;
        hipError_t r = hipStreamEndCapture(e->stream, &g);
        if (r == hipSuccess) r = hipGraphInstantiate(&tg->exec, g, nullptr, nullptr, 0);
        if (g) (void)hipGraphDestroy(g);
        if (r != hipSuccess || !e->bt_dev_valid) {       // nothing has run: undo the host-side step bookkeeping
            (void)hipGetLastError();
            if (tg->exec) { (void)hipGraphExecDestroy(tg->exec); tg->exec = nullptr; }
            e->bt1 = s1; e->bt2 = s2; e->bt_dev_valid = false;
            e->tg_disabled = true;
            return false;
        }
        tg->bs = bs; tg->nb = nb; tg->flags = flags; tg->xs = xs_rows;
    } else {
        // what the launch functions keep track of on the host
        for (long long b = 0; b < nb; ++b) { e->bt1 *= e->cfg.beta1; e->bt2 *= e->cfg.beta2; }
        e->last_nslab = nslab;
        e->w16_dirty = true;
        e->wt_dirty = true;
        e->loss_acc_zeroed = false;
    }
    if (hipGraphLaunch(tg->exec, e->stream) != hipSuccess) {
        // cannot happen after a successful instantiate; if it does the epoch has not run and the state says so
        (void)hipGetLastError();
        e->tg_disabled = true;
        fail(ISK_EHIP, "hipGraphLaunch failed");
        return true;
    }
    return true;
}

// fold_behind: the iteration's last epoch -- the forward weight images of the NEXT iteration are rebuilt right behind the
// loss read-back, i.e. while the host turns around (it waits for the read-back's event, not for the stream)
static void fold_ahead(isk_engine* e);
// the host side of an epoch's end: wait for the event behind its last optimizer step, then read what that step published
static int finish_epoch(isk_engine* e, double* loss, int* status_word) {
    HIPCHK(hipEventSynchronize(e->ev_loss));                     // the shuffle order's copy (pinned staging) precedes it on the stream
    HIPCHK(hipGetLastError());
    const double* acc = e->h_res;
    if (status_word) std::memcpy(status_word, acc + 2, sizeof(int));
    if (acc[1] != 0.0) return fail(ISK_ECOLLAPSED, isk_status_string(ISK_ECOLLAPSED));
    if (loss) *loss = acc[0] / (double)e->N;                     // src/iso.jl:193
    return ISK_OK;
}

// defer: return as soon as the epoch is enqueued; the caller collects its result with finish_epoch (isk_run does so after
// it has launched the NEXT iteration's forward, so the host's turn-around runs under that kernel)
static int train_epoch(isk_engine* e, int32_t minibatch, const int32_t* perm, double* loss, int* status_word = nullptr,
                       bool perm_staged = false, bool fold_behind = false, bool defer = false) {
    const long long N = e->N;
    const int mb = minibatch < 0 ? e->cfg.minibatch : minibatch;
    const long long bs = (mb == 0 || N < mb) ? N : mb;          // src/iso.jl:180
    const long long nb = N / bs;                                 // partial=false drops the tail
    std::vector<int32_t> p;
    if (perm_staged) {
        HIPCHK(hipStreamWaitEvent(e->stream, e->ev_perm, 0));
    } else {
        if (perm) {
            for (long long i = 0; i < N; ++i)
                if (perm[i] < 0 || perm[i] >= N) return fail(ISK_EINVAL, "permutation entry out of range");
        } else {
            p.resize(N);
            shuffle_perm(e, p);
            perm = p.data();
        }
        std::memcpy(e->h_perm, perm, (size_t)N * sizeof(int));      // the previous epoch's copy completed before its final sync
        HIPCHK(hipMemcpyAsync(e->d_perm, e->h_perm, (size_t)N * sizeof(int), hipMemcpyHostToDevice, e->stream));
    }
    if (!e->loss_acc_zeroed) HIPCHK(hipMemsetAsync(e->d_loss_acc, 0, 2 * sizeof(double), e->stream));
    e->loss_acc_zeroed = false;
    {
        ScopedTimer t(e, ISK_T_TRAIN);
        const bool lazy_fold = e->layered && !e->cfg.layernorm && blas_ready(e);
        if (!train_epoch_graph(e, bs, nb)) {
            for (long long b = 0; b < nb; ++b) {
                launch_grads(e, e->d_perm + b * bs, (int)bs, 1.0f / (float)bs, pick_slabs((int)bs));
                launch_opt(e, 1.0f, !lazy_fold || b == nb - 1, false, b == nb - 1);
            }
        }
    }
    // one round trip per epoch: loss, collapse flag and (for isk_run) the target's status word, into pinned memory
    // loss, collapse flag and the target's status word were published into mapped pinned memory by the last opt_step
    double* acc = e->h_res;
    if (nb < 1) {                        // no optimizer step ran (cannot happen: bs <= N), keep the read-back well defined
        HIPCHK(hipMemcpyAsync(acc, e->d_loss_acc, 2 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(acc + 2, e->d_status, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    }
    HIPCHK(hipEventRecord(e->ev_loss, e->stream));
    if (fold_behind) fold_ahead(e);
    if (defer) return ISK_OK;
    return finish_epoch(e, loss, status_word);
}

extern "C" int isk_train_batch(isk_engine* e, const float* target, int32_t minibatch, const int32_t* perm, double* loss) {
    CHK(need_data(e));
    if (target) CHK(upload_target(e, target));
    if (!e->have_target) return fail(ISK_ESTATE, "no target: pass one or call isk_target first");
    return train_epoch(e, minibatch, perm, loss);
}

extern "C" int isk_batch_grads(isk_engine* e, const float* target, const int32_t* idx, int32_t nidx, float* grads, double* loss) {
    CHK(need_data(e));
    if (!idx || !grads || nidx < 1 || nidx > e->N) return fail(ISK_EINVAL, "bad batch");
    for (int i = 0; i < nidx; ++i)
        if (idx[i] < 0 || idx[i] >= e->N) return fail(ISK_EINVAL, "batch index out of range");
    if (target) CHK(upload_target(e, target));
    if (!e->have_target) return fail(ISK_ESTATE, "no target: pass one or call isk_target first");
    HIPCHK(hipMemcpyAsync(e->d_perm, idx, (size_t)nidx * sizeof(int), hipMemcpyHostToDevice, e->stream));
    launch_grads(e, e->d_perm, nidx, 1.0f / (float)nidx, 1, false);
    std::vector<float> g((size_t)e->nparams + 1);
    HIPCHK(hipMemcpyAsync(g.data(), e->d_grad, g.size() * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipGetLastError());
    engine_to_api(e, g.data(), grads);
    if (loss) *loss = (double)g[e->nparams];
    return ISK_OK;
}

extern "C" int isk_set_minibatch(isk_engine* e, int32_t minibatch) {
    if (!e) return fail(ISK_EINVAL, "null engine");
    if (minibatch < 0) return fail(ISK_EINVAL, "negative minibatch");
    e->cfg.minibatch = minibatch;
    return ISK_OK;
}

extern "C" int isk_run(isk_engine* e, int32_t n, int32_t epochs, const int32_t* perms, double* losses) {
    CHK(need_data(e));
    if (n < 0 || epochs < 0) return fail(ISK_EINVAL, "negative iteration count");
    long long k = 0;
    for (int it = 0; it < n; ++it) {
        // launch order = what the GPU can start first: the forward (and the short target chain behind it) goes out BEFORE
        // the host spends ~50 us drawing and staging the epoch's shuffle order -- the draw then runs under the forward
        // instead of in front of it with the GPU idle (the copy is on a stream of its own, the training waits for its event)
        if (e->cfg.featurize_every_iteration) featurize_resident(e);
        {
            const int mb = e->cfg.minibatch;
            const long long bs = (mb == 0 || e->N < mb) ? e->N : mb;
            static const int early = [] { const char* v = getenv("ISK_SIDE_EARLY"); return v ? atoi(v) : 1; }();
            e->side_after_forward = early ? (e->N / bs) * epochs : 0;
            if (early) featurize_early_share(e, e->side_after_forward);
        }
        // the constant-chi check needs the device status word; it is read together with the loss
        const int tr = compute_target(e, false, e->cfg.featurize_every_iteration != 0);
        e->side_after_forward = 0;
        CHK(tr);
        if (epochs > 0) CHK(stage_perm_ahead(e, perms ? perms + k * e->N : nullptr));
        {
            const int mb = e->cfg.minibatch;
            const long long bs = (mb == 0 || e->N < mb) ? e->N : mb;
            featurize_ys_beside_training(e, (e->N / bs) * epochs);          // no-op when the hook already launched it
        }
        for (int ep = 0; ep < epochs; ++ep) {
            double l = 0;
            const int32_t* p = perms ? perms + k * e->N : nullptr;
            // (collecting an iteration's loss only after the NEXT iteration's forward has been launched -- so that the host's
            // turn-around runs under that kernel -- was built and measured: +-0 at minibatch 1000, 5 % SLOWER at minibatch 100)
            int st = 0;
            int r = train_epoch(e, -1, p, &l, &st, ep == 0, ep == epochs - 1);
            if (st) {   // the target's DomainError (constant chi / singular simplex) outranks the loss check
                const int sr = check_status(e);
                if (sr != ISK_OK) return sr;
            }
            if (r != ISK_OK) return r;
            if (losses) losses[k] = l;
            ++k;
        }
        if (epochs == 0) CHK(check_status(e));
    }
    return ISK_OK;
}

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
    CHK(need_data(e));
    if (e->cfg.target_kind != ISK_TARGET_ISA) return fail(ISK_EINVAL, "engine is not configured for TransformISA");
    if (!ks_all_dev || !chi_all_dev || n_global < e->N || first < 0 || first + e->N > n_global) return fail(ISK_EINVAL, "bad gathered arrays");
    const int d = e->nout;
    if (e->isa_work_dp_rows < n_global) {
        HIPCHK(hipStreamSynchronize(e->stream));
        free_dev(e->d_isa_work_dp);
        e->d_isa_work_dp = nullptr;
        HIPCHK(hipMalloc((void**)&e->d_isa_work_dp, (size_t)isk_isa_work_doubles(n_global, d) * sizeof(double)));
        e->isa_work_dp_rows = n_global;
    }
    IsaArgs ia;
    ia.ks = ks_all_dev; ia.chi = chi_all_dev; ia.work = e->d_isa_work_dp; ia.target = e->d_target;
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

extern "C" int isk_isa_solve(const float* ks, const float* chi, int64_t N, int32_t d, int32_t permute, int32_t whitening,
                             double* At, int32_t* rowperm) {
    if (!ks || !chi || !At || !rowperm || N < 1 || d < 2 || d > 8) return fail(ISK_EINVAL, "bad argument");
    isk_engine tmp;
    tmp.cfg.isa_permute = permute; tmp.cfg.isa_whitening = whitening;
    std::vector<float> k(ks, ks + (size_t)N * d), c(chi, chi + (size_t)N * d);
    std::vector<double> A;
    std::vector<int> rp;
    CHK(isa_solve_host(&tmp, k, c, N, d, A, rp, nullptr));
    for (int i = 0; i < d * d; ++i) At[i] = A[i];
    for (int i = 0; i < d; ++i) rowperm[i] = rp[i];
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
    HIPCHK(hipMemsetAsync(e->d_loss_acc, 0, 2 * sizeof(double), e->stream));
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
    launch_opt(e, 1.0f);
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

// ---------------------------------------------------------------------------------------
// native collectives (RCCL, loaded lazily so the library has no link-time dependency on it)
// ---------------------------------------------------------------------------------------
namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;          // optional: the two gathers of the ISA target go out as one group
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int rccl_load() {
    if (g_rccl.lib) return ISK_OK;
    // the copy the host process already uses first (PyTorch ships its own librccl: two different RCCL builds in one
    // process would each keep their own view of the devices and of the IPC state)
    void* h = nullptr;
    if (const char* over = getenv("ISK_RCCL_LIB")) {     // another collective library with the same six entry points (tests/mock_rccl)
        h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
        if (!h) return fail(ISK_EUNSUPPORTED, std::string("cannot load ISK_RCCL_LIB: ") + dlerror());
    }
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(ISK_EUNSUPPORTED, std::string("cannot load librccl.so: ") + dlerror());
    RcclApi a;
    a.lib = h;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
    a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
    if (!a.GroupStart || !a.GroupEnd) a.GroupStart = a.GroupEnd = nullptr;
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.AllGather || !a.GetErrorString)
        return fail(ISK_EUNSUPPORTED, "librccl.so lacks a required symbol");
    g_rccl = a;
    return ISK_OK;
}
}  // namespace

#define RCCLCHK(call)                                                                              \
    do {                                                                                           \
        ncclResult_t _r = (call);                                                                  \
        if (_r != ncclSuccess) return fail(ISK_EHIP, std::string("rccl: ") + g_rccl.GetErrorString(_r)); \
    } while (0)

static void dp_comm_release(isk_engine* e) {
    if (e->comm && g_rccl.CommDestroy) { (void)hipStreamSynchronize(e->stream); (void)g_rccl.CommDestroy(e->comm); }
    e->comm = nullptr;
    e->dp_rank = 0; e->dp_world = 1; e->dp_checked_N = -1;
}

extern "C" int isk_dp_comm_id(uint8_t id[128]) {
    if (!id) return fail(ISK_EINVAL, "null argument");
    CHK(rccl_load());
    ncclUniqueId u;
    RCCLCHK(g_rccl.GetUniqueId(&u));
    static_assert(sizeof(u) == 128, "ncclUniqueId");
    std::memcpy(id, &u, 128);
    return ISK_OK;
}

extern "C" int isk_dp_comm_init(isk_engine* e, const uint8_t id[128], int32_t rank, int32_t world) {
    if (!e || !id || world < 1 || rank < 0 || rank >= world) return fail(ISK_EINVAL, "bad communicator arguments");
    CHK(rccl_load());
    HIPCHK(hipSetDevice(e->cfg.device));
    dp_comm_release(e);
    ncclUniqueId u;
    std::memcpy(&u, id, 128);
    RCCLCHK(g_rccl.CommInitRank(&e->comm, world, u, rank));
    e->dp_rank = rank; e->dp_world = world;
    return ISK_OK;
}

// mm[0] <- -mm[0]: with [-min, max] ONE all-reduce(max) yields the global range
__global__ void negate_first_kernel(float* mm) { if (threadIdx.x == 0 && blockIdx.x == 0) mm[0] = -mm[0]; }

// collectives run when there is more than one rank (ISK_DP_FORCE_COLL=1: also on a single-rank communicator, for tests)
static bool dp_collectives(const isk_engine* e) {
    static const bool force = [] { const char* v = getenv("ISK_DP_FORCE_COLL"); return v && atoi(v) != 0; }();
    return e->comm != nullptr && (e->dp_world > 1 || force);
}

static int dp_target_native(isk_engine* e) {            // the caller has run featurize_resident
    const int d = e->nout;
    const bool coll = dp_collectives(e);
    if (e->cfg.target_kind == ISK_TARGET_SHIFTSCALE) {
        CHK(forward_resident(e, 1, e->cfg.featurize_every_iteration != 0));
        side_featurize_hook(e);
        ScopedTimer t(e, ISK_T_TARGET);
        launch_kmean(e);
        isk_launch_minmax(e->d_ks, e->N, e->d_mm, e->stream);
        if (coll) {
            hipLaunchKernelGGL(negate_first_kernel, dim3(1), dim3(64), 0, e->stream, e->d_mm);
            RCCLCHK(g_rccl.AllReduce(e->d_mm, e->d_mm, 2, ncclFloat, ncclMax, e->comm, e->stream));
            hipLaunchKernelGGL(negate_first_kernel, dim3(1), dim3(64), 0, e->stream, e->d_mm);
        }
        isk_launch_shiftscale(e->d_ks, e->d_mm, e->d_target, e->N, e->d_status, e->stream);   // flags max <= min (also NaN)
        e->have_target = true;
        return ISK_OK;
    }
    if (e->cfg.target_kind != ISK_TARGET_ISA) return fail(ISK_EINVAL, "host-side target transform: no native data-parallel run");
    CHK(forward_resident(e, 3, e->cfg.featurize_every_iteration != 0));
    side_featurize_hook(e);
    ScopedTimer t(e, ISK_T_TARGET);
    launch_kmean(e);
    const float* ks = e->d_ks;
    const float* chi = e->d_out + e->N * e->K * d;
    long long n_global = e->N, first = 0;
    if (coll) {
        n_global = e->N * e->dp_world; first = e->N * e->dp_rank;
        if (e->gather_rows < n_global) {
            HIPCHK(hipStreamSynchronize(e->stream));
            free_dev(e->d_gks); free_dev(e->d_gchi);
            e->d_gks = e->d_gchi = nullptr;
            HIPCHK(hipMalloc((void**)&e->d_gks, (size_t)n_global * d * sizeof(float)));
            HIPCHK(hipMalloc((void**)&e->d_gchi, (size_t)n_global * d * sizeof(float)));
            e->gather_rows = n_global;
        }
        // one group: RCCL aggregates the two small gathers into a single launch (each is latency, not bytes)
        if (g_rccl.GroupStart) RCCLCHK(g_rccl.GroupStart());
        const ncclResult_t r1 = g_rccl.AllGather(ks, e->d_gks, (size_t)e->N * d, ncclFloat, e->comm, e->stream);
        const ncclResult_t r2 = g_rccl.AllGather(chi, e->d_gchi, (size_t)e->N * d, ncclFloat, e->comm, e->stream);
        const ncclResult_t r3 = g_rccl.GroupEnd ? g_rccl.GroupEnd() : ncclSuccess;      // always closed, whatever r1 / r2 say
        RCCLCHK(r1); RCCLCHK(r2); RCCLCHK(r3);
        ks = e->d_gks; chi = e->d_gchi;
    }
    return isk_dp_isa_dev(e, ks, chi, n_global, first);
}

extern "C" int isk_dp_run(isk_engine* e, int32_t n, int32_t epochs, int32_t local_batch, int32_t nbatches, int64_t global_batch,
                          const int32_t* perms, double* losses) {
    CHK(need_data(e));
    if (n < 0 || epochs < 0) return fail(ISK_EINVAL, "negative iteration count");
    if (local_batch < 1 || nbatches < 1 || (long long)local_batch * nbatches > e->N || global_batch < local_batch)
        return fail(ISK_EINVAL, "bad batch decomposition");
    // every argument check comes BEFORE the first collective: a rank that returned early from inside the loop would
    // leave the others blocked in an all-reduce
    if (perms)
        for (long long i = 0; i < (long long)n * epochs * e->N; ++i)
            if (perms[i] < 0 || perms[i] >= e->N) return fail(ISK_EINVAL, "permutation entry out of range");
    if (e->cfg.target_kind != ISK_TARGET_SHIFTSCALE && e->cfg.target_kind != ISK_TARGET_ISA)
        return fail(ISK_EINVAL, "host-side target transform: no native data-parallel run");
    const bool coll = dp_collectives(e);
    if (coll && e->dp_checked_N != e->N) {          // equal shards: the gathered ISA arrays and the step count assume it (checked once)
        long long* d_n = nullptr;
        HIPCHK(hipMalloc((void**)&d_n, 2 * sizeof(long long)));
        const long long hn[2] = {e->N, -e->N};
        HIPCHK(hipMemcpyAsync(d_n, hn, sizeof(hn), hipMemcpyHostToDevice, e->stream));
        ncclResult_t r = g_rccl.AllReduce(d_n, d_n, 2, ncclInt64, ncclMax, e->comm, e->stream);
        long long gn[2] = {0, 0};
        if (r == ncclSuccess) (void)hipMemcpyAsync(gn, d_n, sizeof(gn), hipMemcpyDeviceToHost, e->stream);
        (void)hipStreamSynchronize(e->stream);
        (void)hipFree(d_n);
        if (r != ncclSuccess) return fail(ISK_EHIP, std::string("rccl: ") + g_rccl.GetErrorString(r));
        if (gn[0] != e->N || gn[1] != -e->N) return fail(ISK_EINVAL, "isk_dp_run needs equal shards on every rank");
        e->dp_checked_N = e->N;
    }
    // with real collectives keep ONE gradient slab: the all-reduce is latency/link-bound, half the bytes beat the
    // few microseconds the second row slab saves in train_grad1
    const int nslab = coll ? 1 : pick_slabs(local_batch);
    const size_t red_count = (size_t)(nslab - 1) * (e->nparams + 2) + (size_t)e->nparams + 1;
    long long k = 0;
    for (int it = 0; it < n; ++it) {
        if (e->cfg.featurize_every_iteration) featurize_resident(e);      // first: see isk_run
        e->side_after_forward = (long long)nbatches * epochs;
        featurize_early_share(e, e->side_after_forward);
        const int tr = dp_target_native(e);
        e->side_after_forward = 0;
        CHK(tr);
        // the shuffle order (drawn here from the engine's seeded generator when the host passes none: ~50 us of host
        // work) is staged while the forward runs
        if (epochs > 0) CHK(stage_perm_ahead(e, perms ? perms + k * e->N : nullptr));
        featurize_ys_beside_training(e, (long long)nbatches * epochs);
        for (int ep = 0; ep < epochs; ++ep) {
            if (ep == 0) {
                HIPCHK(hipStreamWaitEvent(e->stream, e->ev_perm, 0));
            } else {
                std::vector<int32_t> drawn;
                const int32_t* p = perms ? perms + k * e->N : nullptr;
                if (!p) { drawn.resize(e->N); shuffle_perm(e, drawn); p = drawn.data(); }
                std::memcpy(e->h_perm, p, (size_t)e->N * sizeof(int));     // validated before the first collective
                HIPCHK(hipMemcpyAsync(e->d_perm, e->h_perm, (size_t)e->N * sizeof(int), hipMemcpyHostToDevice, e->stream));
            }
            if (!e->loss_acc_zeroed) HIPCHK(hipMemsetAsync(e->d_loss_acc, 0, 2 * sizeof(double), e->stream));
            e->loss_acc_zeroed = false;
            {
                ScopedTimer t(e, ISK_T_TRAIN);
                for (int b = 0; b < nbatches; ++b) {
                    launch_grads(e, e->d_perm + (long long)b * local_batch, local_batch, 1.0f / (float)global_batch, nslab);
                    if (coll) RCCLCHK(g_rccl.AllReduce(e->d_grad, e->d_grad, red_count, ncclFloat, ncclSum, e->comm, e->stream));
                    launch_opt(e, 1.0f, true, false, b == nbatches - 1);
                }
            }
            double* acc = e->h_res;          // published by the last opt_step (mapped pinned memory)
            HIPCHK(hipEventRecord(e->ev_loss, e->stream));
            if (ep == epochs - 1) fold_ahead(e);         // the next forward's weight images, while the host turns around (see train_epoch)
            HIPCHK(hipEventSynchronize(e->ev_loss));
            HIPCHK(hipGetLastError());
            int st = 0;
            std::memcpy(&st, acc + 2, sizeof(int));
            if (st) {   // the target's DomainError outranks the loss check; every rank sees the same status (same gathered data)
                const int sr = check_status(e);
                if (sr != ISK_OK) return sr;
            }
            if (acc[1] != 0.0) return fail(ISK_ECOLLAPSED, isk_status_string(ISK_ECOLLAPSED));
            if (losses) losses[k] = acc[0];
            ++k;
        }
        if (epochs == 0) CHK(check_status(e));
    }
    return ISK_OK;
}

// ---------------------------------------------------------------------------------------
// instrumentation
// ---------------------------------------------------------------------------------------
extern "C" int isk_timers_enable(isk_engine* e, int32_t on) {
    if (!e) return fail(ISK_EINVAL, "null engine");
    e->timers_on = on != 0;
    return ISK_OK;
}
extern "C" int isk_timers_read(isk_engine* e, double total_ms[ISK_T_COUNT], int64_t count[ISK_T_COUNT]) {
    if (!e) return fail(ISK_EINVAL, "null engine");
    drain_timers(e);
    for (int i = 0; i < ISK_T_COUNT; ++i) { if (total_ms) total_ms[i] = e->t_ms[i]; if (count) count[i] = e->t_cnt[i]; }
    return ISK_OK;
}
extern "C" int isk_timers_reset(isk_engine* e) {
    if (!e) return fail(ISK_EINVAL, "null engine");
    drain_timers(e);
    for (int i = 0; i < ISK_T_COUNT; ++i) { e->t_ms[i] = 0; e->t_cnt[i] = 0; }
    return ISK_OK;
}
