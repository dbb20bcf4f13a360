// Host side of the MI355X ISOKANN engine: owns device memory, sequences the kernels of
// kernels_fwd.hip / kernels_fwd_bf16.hip / kernels_train.hip / kernels_isa.hip / kernels_layered.hip on the engine's
// streams (main, side: the next iteration's featurizer, copy: shuffle orders), implements the C ABI of
// include/isokann_hip.h.  Nothing of an isk_run iteration runs on the host: the Float64 inner-simplex (ISA) target is
// a device kernel chain (the reference pulls chi and ks to the CPU, src/isotarget.jl:81-107; the host twin
// isa_solve_host below serves the host-driven data-parallel hooks and isk_isa_solve).
#include "engine.hpp"

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
static thread_local std::string g_err;

int isk_fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

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
        case ISK_EPEER: return "a peer rank of the data-parallel group reported an error";
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
// padded input width of layer l: the feature rows are padded to 32; hidden activations to 16 on the fused path
// and to 32 on the layered path (the split-bf16 GEMM contracts 32 at a time)
int layer_kp(const isk_engine* e, int l) {
    if (l == 0) return e->Fp;
    return e->layered ? isk_pad32(e->cfg.widths[l]) : isk_pad16(e->cfg.widths[l]);
}

void free_dev(void* p) { if (p) (void)hipFree(p); }

static void free_data(isk_engine* e) {
    if (e->side) (void)hipStreamSynchronize(e->side);
    e->ys_ahead = false;
    for (auto& g : e->tg) { if (g.exec) (void)hipGraphExecDestroy(g.exec); g = isk_engine::TrainGraph(); }
    free_dev(e->d_in_ys); free_dev(e->d_xs32_alt); e->d_xs32_alt = nullptr; free_dev(e->d_xhat); free_dev(e->d_xhat_alt); free_dev(e->d_out); free_dev(e->d_ks);
    free_dev(e->d_xscale); free_dev(e->d_xscale_alt); free_dev(e->d_xs32);
    e->d_xscale = e->d_xscale_alt = e->d_xs32 = nullptr;
    free_dev(e->d_target); free_dev(e->d_kw); e->d_kw = nullptr; free_dev(e->d_z1); free_dev(e->d_tail_cnt); e->d_tail_cnt = nullptr; free_dev(e->d_delta1); free_dev(e->d_part); free_dev(e->d_perm); free_dev(e->d_isa_work);
    e->d_isa_work = nullptr;
    free_dev(e->d_ks64); e->d_ks64 = nullptr;
    free_dev(e->d_actA); free_dev(e->d_actB); e->d_actA = e->d_actB = nullptr;
    free_dev(e->dbg_st); free_dev(e->dbg_rows); e->dbg_st = nullptr; e->dbg_rows = nullptr;      // (dbg_rows is sized by the data)
    bf16_train_release(e);
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
void shuffle_perm(isk_engine* e, std::vector<int32_t>& p) {
    for (size_t i = 0; i < p.size(); ++i) p[i] = (int32_t)i;
    for (size_t i = p.size(); i > 1; --i) {
        const size_t j = (size_t)(xoshiro(e->rng) % i);
        std::swap(p[i - 1], p[j]);
    }
}

// ---- argument builders ----------------------------------------------------------------
int act_of_layer(const isk_engine* e, int l) { return l == e->L - 1 ? e->cfg.lastactivation : e->cfg.activation; }

MlpFwdArgs fwd_args(const isk_engine* e, const float* X, long long S, float* out) {
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
    a.sched_skew = e->opt.bf16_skew;
    a.waves_per_simd = e->opt.fwd_waves_per_simd;
    return a;
}

TrainShape train_shape(const isk_engine* e) {
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

long long z1_rows(long long B) { return ((B + 31) / 32) * 32; }
TrainBuffers train_buffers(const isk_engine* e) {
    TrainBuffers b;
    std::memset(&b, 0, sizeof(b));
    b.beta1 = e->cfg.beta1; b.beta2 = e->cfg.beta2;
    b.xhat_xs = e->d_xs32;      // a buffer of its own: d_xhat / d_xhat_alt swap every iteration (and hold f16 planes in forward_mode 3)
    b.target = e->d_target; b.lossw = e->d_lossw; b.theta = e->d_theta;
    b.W1t = e->d_Wt[0]; b.bias1 = e->d_bias[0];
    b.a1 = e->d_z1; b.delta1 = e->d_delta1; b.part = e->d_part; b.grad = e->d_grad;
    b.loss_acc = e->d_loss_acc; b.Fp = e->Fp;
    b.rank_err = e->d_rank_err;
    b.tail_generic = e->opt.tail_narrow ? 0 : 1;
    return b;
}

void launch_fold(isk_engine* e) {
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
    e->b16.wt_dirty = e->b16.wf_dirty = true;
}
// fused path: the training steps read theta itself, the forward weight images are rebuilt when a forward needs them
void ensure_folded(isk_engine* e) {
    if (e->wt_dirty) launch_fold(e);
}

// ---- API (Flux) order <-> engine order, on the host ------------------------------------
static inline long long w1_native_index(const isk_engine* e, int o, int k) {
    const int nt = isk_pad16(e->cfg.widths[1]) / 16;
    return e->off_W[0] + (((long long)(k >> 4) * nt + (o >> 4)) * 64 + (((k & 15) >> 2) << 4 | (o & 15))) * 4 + (k & 3);
}
void api_to_engine(const isk_engine* e, const float* a, std::vector<float>& out) {
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
void engine_to_api(const isk_engine* e, const float* in, float* a) {
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
    if (cfg->target_kind == ISK_TARGET_ISA && nout > ISK_ISA_HOST_MAXD) return fail(ISK_EUNSUPPORTED, "TransformISA: at most 64 outputs");
    for (int l = 1; l <= cfg->n_layers; ++l)
        if (cfg->widths[l] > 16384) return fail(ISK_EUNSUPPORTED, "layer width > 16384");
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
    {
        // rows the register-resident featurizer does not cover (> 2048 features, > 682 atoms) go through the LDS-stash kernel: four
        // waves, each with one sample's coordinates and feature row in LDS
        const long long Dp = cfg->featurizer != ISK_FEAT_IDENTITY ? ((Din + 3) & ~3) : 0;
        const bool reg = isk_pad32(F) <= 2048 && (cfg->featurizer == ISK_FEAT_IDENTITY || Din <= 2048);
        if (!reg && (size_t)4 * (size_t)(Dp + isk_pad32(F)) * sizeof(float) > (size_t)160 * 1024)
            return fail(ISK_EUNSUPPORTED, "features + coordinates per sample > 10240 (beyond 2048 features / 682 atoms the featurizer keeps a sample in LDS)");
    }
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(ISK_EINVAL, "device ordinal out of range");
    HIPCHK(hipSetDevice(cfg->device));

    isk_engine* e = new isk_engine();
    e->cfg = *cfg;
    e->cfg.pairs = nullptr;
    e->L = cfg->n_layers; e->F = F; e->Fp = isk_pad32(F); e->Din = Din; e->nout = nout;
    engine_options_init(e);
    // structural choices are made here, once: the environment of the CREATING call may override the two config fields that
    // select them (test sweeps run the whole suite in another forward arithmetic / on the layered path); nothing is latched
    if (const char* fm = getenv("ISK_FWD_MODE")) e->cfg.forward_mode = atoi(fm);
    if (getenv("ISK_FORCE_LAYERED")) e->cfg.force_layered = 1;
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
        if (e->cfg.force_layered) fits = false;
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
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess) e->ncu = ncu;
    }

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
        // behind the epoch's last opt_step, whose stores into mapped host memory the host reads next: release to system scope
        hipEventCreateWithFlags(&e->ev_loss, hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess) {
        isk_destroy(e);
        return fail(ISK_EHIP, "side stream / event creation failed");
    }
    // coherent (fine-grained) mapping requested explicitly: the read-back must not depend on HIP_HOST_COHERENT's default
    if (hipHostMalloc((void**)&e->h_res, 4 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void**)&e->d_hres, e->h_res, 0) != hipSuccess) { isk_destroy(e); return fail(ISK_ENOMEM, "hipHostMalloc (mapped)"); }
    const size_t pb = (size_t)(e->nparams + 2) * sizeof(float);
    ALLOC(e->d_theta, pb); ALLOC(e->d_grad, pb * ISK_MAX_SLABS); ALLOC(e->d_m, pb); ALLOC(e->d_v, pb);
    ALLOC(e->d_loss_acc, 4 * sizeof(double)); ALLOC(e->d_mm, 2 * sizeof(float));
    e->d_status = (int*)(e->d_loss_acc + 2);         // the target's status word lives behind the loss accumulator (opt_step publishes both)
    ALLOC(e->d_bt, 4 * sizeof(double));
    ALLOC(e->d_rank_err, sizeof(int));
    (void)hipMemset(e->d_rank_err, 0, sizeof(int));
    if (hipHostMalloc((void**)&e->h_flag, sizeof(int), hipHostMallocDefault) != hipSuccess) { isk_destroy(e); return fail(ISK_ENOMEM, "hipHostMalloc"); }
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
        ALLOC(e->d_zero, (size_t)16384 * sizeof(float));                 // a zero bias as wide as the widest layer allowed
        (void)hipMemset(e->d_zero, 0, (size_t)16384 * sizeof(float));
        if (bf16_weight_images_alloc(e) != ISK_OK) { isk_destroy(e); return fail(ISK_ENOMEM, "hipMalloc: bf16 weight images"); }
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
    std::memset(e->h_res, 0, 4 * sizeof(double));
    std::vector<float> ones(nout, 1.0f);
    (void)hipMemcpy(e->d_lossw, ones.data(), nout * sizeof(float), hipMemcpyHostToDevice);
    launch_fold(e);
    (void)hipStreamSynchronize(e->stream);
    *out = e;
    return ISK_OK;
}


extern "C" void isk_destroy(isk_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->cfg.device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->side) (void)hipStreamSynchronize(e->side);
    if (e->copy) (void)hipStreamSynchronize(e->copy);
    free_data(e);
    if (e->d_grad_base) e->d_grad = e->d_grad_base;                  // (the peer-reduce run alternates d_grad between two slabs of its allocation)
    free_dev(e->d_theta); free_dev(e->d_grad); free_dev(e->d_m); free_dev(e->d_v); free_dev(e->d_bt);
    free_dev(e->d_ep_zp); free_dev(e->d_ep_lnpart); free_dev(e->d_ep_cnt);
    if (e->comm_stream) { (void)hipStreamSynchronize(e->comm_stream); (void)hipStreamDestroy(e->comm_stream); }
    for (auto& ev : e->ev_grads) if (ev) (void)hipEventDestroy(ev);
    if (e->grp && e->dp_rank == 0) delete e->grp;                    // the group object belongs to its first engine
    if (e->ev_b1_ready) (void)hipEventDestroy(e->ev_b1_ready);
    if (e->ev_b1_done) (void)hipEventDestroy(e->ev_b1_done);
    free_dev(e->d_loss_acc); free_dev(e->d_mm); free_dev(e->d_lossw);
    free_dev(e->d_At); free_dev(e->d_rowperm); free_dev(e->d_pairs); free_dev(e->d_packed); free_dev(e->d_isa_work_dp);
    for (int l = 0; l < ISK_MAX_LAYERS; ++l) { free_dev(e->d_Wt[l]); free_dev(e->d_bias[l]); }
    free_dev(e->d_W16); free_dev(e->d_wmax); free_dev(e->d_wscale);
    for (int l = 0; l < ISK_MAX_LAYERS; ++l) { free_dev(e->d_W16L[l]); free_dev(e->d_WtT[l]); }
    free_dev(e->d_G0); free_dev(e->d_D0); free_dev(e->d_zero);
    for (int l = 0; l < ISK_MAX_LAYERS; ++l) { free_dev(e->b16.Wt[l]); free_dev(e->b16.Wf[l]); }
    dp_comm_release(e);
    free_dev(e->d_gks); free_dev(e->d_gchi); free_dev(e->d_gks64);
    drain_timers(e);
    for (auto& ep : e->ev_free) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    if (e->ev_fwd_done) (void)hipEventDestroy(e->ev_fwd_done);
    if (e->ev_ys_ready) (void)hipEventDestroy(e->ev_ys_ready);
    if (e->ev_perm) (void)hipEventDestroy(e->ev_perm);
    if (e->ev_loss) (void)hipEventDestroy(e->ev_loss);
    if (e->h_res) (void)hipHostFree(e->h_res);
    if (e->h_flag) (void)hipHostFree(e->h_flag);
    free_dev(e->d_rank_err);
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
FeatArgs feat_args(const isk_engine* e) {
    FeatArgs a = {};
    a.pairs = e->d_pairs; a.packed_pairs = e->d_packed; a.D = e->Din; a.F = e->F; a.Fp = e->Fp;
    a.normalize = e->cfg.layernorm; a.ln_mode = e->cfg.ln_mode; a.eps = e->cfg.ln_eps;
    a.bg_wg_per_cu = e->opt.feat_wg_per_cu; a.bg_wgs_total = e->opt.feat_bg_wgs; a.bg_pf = e->opt.feat_pf;
    a.dbg_copy = e->opt.debug_feat_copy;
    a.out_bf16 = e->xh16; a.ld16 = isk_ld16(e->Fp);
    return a;
}

// Featurizes S rows into rows [row0, row0 + S) of the feature buffer `xhat` (its BASE).  xscale != nullptr: split-f16 planes
// (forward_mode 3; unit-blocked rows, isk_plane_off) with the inverse row scales at xscale[row0 ..]; xs32: the rows from
// local row x32_first on in f32 as well
static void featurize_rows(isk_engine* e, const float* in, long long S, float* xhat, long long row0, hipStream_t st = nullptr,
                           float* xscale = nullptr, float* xs32 = nullptr, long long x32_first = 0, int bg_wgs = 1) {
    FeatArgs a = feat_args(e);
    a.in = in; a.S = S;
    a.background = (st != nullptr) ? bg_wgs : 0;       // 1: the default throttle; > 1: that many workgroups in total
    a.planes = xscale != nullptr;
    if (a.planes) { a.xhat = xhat; a.row0 = row0; a.xscale = xscale + row0; }
    else if (e->xh16) a.xhat = (float*)((unsigned short*)xhat + row0 * isk_ld16(e->Fp));      // bf16 rows
    else a.xhat = xhat + row0 * e->Fp;
    a.xhat32 = xs32; a.x32_first = x32_first;
    isk_launch_featurize(a, st ? st : e->stream);
}

// per-iteration featurization.  ys (the 95 % share) may already have been produced on the side
// stream, into the other buffer of the pair, while the previous iteration's MFMA-bound forward was
// running; then the main stream only waits for it and swaps buffers.
void featurize_resident(isk_engine* e) {
    if (e->ys_ahead) {
        // the xs rows of the other buffer do not depend on the side stream: featurize them BEFORE waiting for it, so
        // the (un-throttled) xs launch runs under the tail of the throttled ys featurization instead of after it
        // the side-stream launch covered the ys AND the xs rows (one launch over the contiguous input slab), the f32 xs
        // copy for the training kernels included: swap the three buffer pairs
        if (!e->opt.debug_feat_copy) {                 // (diagnostic stream instead of the featurizer: the other buffers hold no features)
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
static int overlap_mode(const isk_engine* e) { return e->opt.overlap_featurize; }

// Called right after this iteration's forward has been enqueued on the main stream: featurize ys for
// the NEXT iteration on the side stream into the other buffer.  The featurizer is HBM-bound, the
// forward f32-MFMA-bound, so the two share the chip.  `ev_main_point` (recorded before the forward)
// orders the write after every earlier reader of that buffer.
static void featurize_ys_ahead(isk_engine* e) {
    // Measured on MI355X (round 1): co-running the HBM-bound featurizer with the MFMA-bound forward
    // slows the forward from 0.83 to 1.5 ms (the featurizer's streaming evicts the L2-resident
    // weights and saturates the memory queues), a net loss against running them back to back.
    // Kept behind ISK_OVERLAP_FEATURIZE=1 for further tuning.
    if (overlap_mode(e) != 1 || !e->d_xhat_alt) return;
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
void featurize_ys_beside_training(isk_engine* e, long long train_launch_steps) {
    // only worth it when the training phase is a long chain of small launches: with a handful of big steps the
    // throttled featurizer would outlast them and become the critical path (measured at minibatch = 0)
    if (train_launch_steps < e->opt.side_min_steps) return;
    if (overlap_mode(e) != 2 || !e->cfg.featurize_every_iteration || !e->d_xhat_alt || e->ys_ahead) return;
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
void featurize_early_share(isk_engine* e, long long train_launch_steps) {
    const int pct = e->opt.feat_early_pct, wgs = e->opt.feat_early_wgs > 1 ? e->opt.feat_early_wgs : 128;
    e->early_rows = 0;
    if (pct <= 0 || train_launch_steps < e->opt.side_min_steps) return;
    if (overlap_mode(e) != 2 || !e->cfg.featurize_every_iteration || !e->d_xhat_alt || e->ys_ahead) return;
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
void side_featurize_hook(isk_engine* e) {
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
    {
        // wide networks, plain-bf16 forward: the featurizer writes the Koopman rows as bf16, exactly the A operand of the fused
        // forward (half the feature bytes written and read); decided here, with the buffers (option wide_fused: next set_data)
        FeatArgs fa = {};
        fa.pairs = e->d_pairs; fa.packed_pairs = e->d_packed; fa.D = e->Din; fa.F = e->F; fa.Fp = e->Fp;
        e->xh16 = e->layered && bf16_planes(e) == 1 && e->opt.wide_fused && wide_forward_shape_ok(e) && isk_featurize_reg_ok(fa);
    }
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
        const size_t cw = (size_t)isk_train_fwdtail_counters(N);
        HIPCHK(hipMalloc((void**)&e->d_tail_cnt, cw * sizeof(unsigned)));
        HIPCHK(hipMemsetAsync(e->d_tail_cnt, 0, cw * sizeof(unsigned), e->stream));
    }
    HIPCHK(hipMalloc((void**)&e->d_delta1, (size_t)Bpad * H1p * sizeof(float)));
    const TrainShape ts = train_shape(e);
    HIPCHK(hipMalloc((void**)&e->d_part, ((size_t)isk_train_part_floats(ts, (int)N) + 4) * sizeof(float)));      // (+4: kernels_train_chain.hip reads float4s that may end 3 floats past the last partial)
    HIPCHK(hipMalloc((void**)&e->d_perm, (size_t)N * sizeof(int)));
    HIPCHK(hipHostMalloc((void**)&e->h_perm, (size_t)N * sizeof(int), hipHostMallocDefault));
    if (e->cfg.target_kind == ISK_TARGET_ISA && e->nout <= ISK_ISA_DEV_MAXD) {
        HIPCHK(hipMalloc((void**)&e->d_isa_work, (size_t)isk_isa_work_doubles(N, e->nout) * sizeof(double)));
        HIPCHK(hipMalloc((void**)&e->d_ks64, (size_t)N * e->nout * sizeof(double)));       // K chi unrounded, for the Float64 solve
    }
    if (e->layered) {
        int mh = 32;
        for (int l = 1; l < e->L; ++l) mh = std::max(mh, isk_pad32(e->cfg.widths[l]));
        if (!e->xh16) {                  // (the fused wide forward keeps the activations in LDS)
            HIPCHK(hipMalloc((void**)&e->d_actA, (size_t)S * mh * sizeof(float)));
            HIPCHK(hipMalloc((void**)&e->d_actB, (size_t)S * mh * sizeof(float)));
            HIPCHK(hipMemsetAsync(e->d_actA, 0, (size_t)S * mh * sizeof(float), e->stream));
            HIPCHK(hipMemsetAsync(e->d_actB, 0, (size_t)S * mh * sizeof(float), e->stream));
        }
        for (int l = 0; l < e->L; ++l) {
            const size_t nb = (size_t)Bpad * isk_pad16(e->cfg.widths[l + 1]) * sizeof(float);
            HIPCHK(hipMalloc((void**)&e->d_lz[l], nb));
            HIPCHK(hipMalloc((void**)&e->d_la[l], nb));
            HIPCHK(hipMalloc((void**)&e->d_ld[l], nb));
            HIPCHK(hipMemsetAsync(e->d_lz[l], 0, nb, e->stream));
            HIPCHK(hipMemsetAsync(e->d_la[l], 0, nb, e->stream));
            HIPCHK(hipMemsetAsync(e->d_ld[l], 0, nb, e->stream));
        }
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

// addcoords! / the cutoff window of run_kde! (src/iso.jl:238,288-290): the resident start points stay on the device
extern "C" int isk_append_data(isk_engine* e, const float* xs, const float* ys, int64_t D, int64_t K, int64_t n_add, int64_t drop_first) {
    CHK(need_data(e));
    if (n_add < 0 || drop_first < 0 || drop_first > e->N) return fail(ISK_EINVAL, "bad row counts");
    if (n_add > 0 && (!xs || !ys)) return fail(ISK_EINVAL, "null argument");
    if (D != e->D || K != e->K) return fail(ISK_EINVAL, "D and K of the new samples must match the resident data");
    const int64_t keep = e->N - drop_first, Nn = keep + n_add;
    if (Nn < 1) return fail(ISK_EINVAL, "no start point would be left");
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->side) HIPCHK(hipStreamSynchronize(e->side));
    TmpBuf txs, tys;
    HIPCHK(hipMalloc(&txs.p, (size_t)Nn * D * sizeof(float)));
    HIPCHK(hipMalloc(&tys.p, (size_t)Nn * K * D * sizeof(float)));
    float* nxs = (float*)txs.p;
    float* nys = (float*)tys.p;
    // N is the slowest axis of xs (D, N) and ys (D, K, N): the kept start points are one contiguous run of each
    HIPCHK(hipMemcpyAsync(nxs, e->d_in_xs + (size_t)drop_first * D, (size_t)keep * D * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(nys, e->d_in_ys + (size_t)drop_first * K * D, (size_t)keep * K * D * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    if (n_add > 0) {
        HIPCHK(hipMemcpyAsync(nxs + (size_t)keep * D, xs, (size_t)n_add * D * sizeof(float), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(nys + (size_t)keep * K * D, ys, (size_t)n_add * K * D * sizeof(float), hipMemcpyHostToDevice, e->stream));
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    return set_data_impl(e, nxs, nys, D, K, Nn, hipMemcpyDeviceToDevice);          // (frees the old buffers, re-featurizes on the device)
}

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
// resident forward, koopman, target
// ---------------------------------------------------------------------------------------
// fused networks: refresh the bf16 weight planes now (launch_forward would do it lazily, i.e. behind whatever the
// stream is about to wait for)
void fold_ahead(isk_engine* e) {
    if (e->layered && e->xh16) { bf16_weight_images_ensure(e, false, true); return; }     // the fused wide forward's own images
    ensure_folded(e);
    const int planes = bf16_planes(e);
    if (e->layered || !(planes || e->planes) || !e->w16_dirty) return;
    if (e->planes) isk_launch_fold_f16(e->d_Wt[0], e->d_W16, e->Fp, isk_pad16(e->cfg.widths[1]) / 16, e->d_wmax + e->wmax_last, e->d_wscale, e->stream);
    else isk_launch_fold_bf16(e->d_Wt[0], e->d_W16, e->Fp, isk_pad16(e->cfg.widths[1]) / 16, planes, e->stream);
    e->w16_dirty = false;
}
// the forward over feature rows: f32 MFMA or split-bf16 (cfg.forward_mode)
int launch_forward(isk_engine* e, const MlpFwdArgs& f) {
    if (e->layered && e->xh16) {                 // fused wide kernel: no activation buffers, its own weight images of theta
        if (e->cfg.layernorm) ensure_folded(e);  //   (with LayerNorm its first bias is the folded one)
        return forward_layered(e, f.X, f.S, f.out, nullptr, nullptr);
    }
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

int need_data(isk_engine* e) {
    if (!e) return fail(ISK_EINVAL, "null engine");
    if (e->N == 0) return fail(ISK_ESTATE, "no data: call isk_set_data first");
    HIPCHK(hipSetDevice(e->cfg.device));
    return ISK_OK;
}

// which: 1 = ys rows, 2 = xs rows, 3 = both (single launch over the contiguous slab)
int forward_resident(isk_engine* e, int which, bool ahead) {
    const long long nys = e->N * e->K;
    const float* X = e->d_xhat;
    float* out = e->d_out;
    long long S = nys + e->N;
    int tid = ISK_T_FWD_YS;
    if (which == 1) S = nys;
    if (which == 2) {
        if (e->xh16) X = (const float*)((const unsigned short*)X + nys * isk_ld16(e->Fp));      // bf16 rows
        else if (!e->planes) X += nys * e->Fp;
        out += nys * e->nout; S = e->N; tid = ISK_T_FWD_XS;
    }
    if (ahead) (void)hipEventRecord(e->ev_fwd_done, e->stream);
    {
        ScopedTimer t(e, tid);
        MlpFwdArgs f = fwd_args(e, X, S, out);
        if (e->planes) { f.xscale = e->d_xscale + (which == 2 ? nys : 0); f.x_row0 = which == 2 ? nys : 0; }   // unit-blocked rows: base + row index
        // diagnostic only (ISK_DEBUG_STAMPS=1 with a -DISK_STAMP build of the bf16x3 kernel): phase cycle sums
        const bool stamps = e->opt.debug_stamps != 0;
        unsigned long long*& dbg_st = e->dbg_st;           // per engine: engines of one process sit on different devices
        if (stamps) {
            if (!dbg_st) (void)hipMalloc((void**)&dbg_st, (size_t)4096 * 12 * 8);
            (void)hipMemsetAsync(dbg_st, 0, (size_t)4096 * 12 * 8, e->stream);
            f.dbg = dbg_st;
        }
        // diagnostic only (ISK_DEBUG_ROWMOD=m): every row reads row (r mod m), i.e. a cache-resident sample stream
        const long long rowmod = e->opt.debug_rowmod;
        int*& dbg_rows = e->dbg_rows;
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

// the epoch accumulator [loss sum, collapsed flag, (target status word), peer flag]: everything but the status word
hipError_t zero_loss_acc(isk_engine* e) {
    hipError_t r = hipMemsetAsync(e->d_loss_acc, 0, 2 * sizeof(double), e->stream);
    if (r == hipSuccess) r = hipMemsetAsync(e->d_loss_acc + 3, 0, sizeof(double), e->stream);
    return r;
}

void launch_kmean(isk_engine* e) {
    KmeanArgs k;
    k.out_ys = e->d_out; k.ks = e->d_ks; k.ks64 = e->d_ks64; k.w = e->d_kw; k.N = e->N; k.K = (int)e->K; k.nout = e->nout;
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
extern "C" int isk_koopman_f64(isk_engine* e, double* out) {
    CHK(need_data(e));
    if (!out) return fail(ISK_EINVAL, "null argument");
    CHK(forward_resident(e, 1));
    launch_kmean(e);
    const size_t n = (size_t)e->N * e->nout;
    if (e->d_ks64) {
        HIPCHK(hipMemcpyAsync(out, e->d_ks64, n * sizeof(double), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
    } else {
        std::vector<float> k32(n);
        HIPCHK(hipMemcpyAsync(k32.data(), e->d_ks, n * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        for (size_t i = 0; i < n; ++i) out[i] = (double)k32[i];
    }
    HIPCHK(hipGetLastError());
    return ISK_OK;
}
int check_status(isk_engine* e) {
    int st = 0;
    HIPCHK(hipMemcpyAsync(&st, e->d_status, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (st) HIPCHK(hipMemsetAsync(e->d_status, 0, sizeof(int), e->stream));
    if (st & 1) return fail(ISK_ECONSTANT_CHI, isk_status_string(ISK_ECONSTANT_CHI));
    if (st & 2) return fail(ISK_ESINGULAR, isk_status_string(ISK_ESINGULAR));
    return ISK_OK;
}

// computes the target on the device-resident data; leaves target + loss weights on the device
int compute_target(isk_engine* e, bool check, bool ahead) {
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
    if (d > ISK_ISA_DEV_MAXD) {
        // more outputs than the device chain is built for (its fixperm walks d! permutations): K chi and chi come to the host, the
        // Float64 solve runs there with fixperm as an assignment problem, target and loss weights go back
        const size_t nf = (size_t)e->N * d;
        std::vector<float> ks(nf), chi(nf);
        HIPCHK(hipMemcpyAsync(ks.data(), e->d_ks, nf * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(chi.data(), e->d_out + e->N * e->K * d, nf * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        std::vector<double> At, tgt;
        std::vector<int> rp;
        CHK(isa_solve_host(e, ks, chi, e->N, d, At, rp, &tgt));
        std::vector<float> tf(nf), w;
        for (size_t i = 0; i < nf; ++i) tf[i] = (float)tgt[i];
        loss_weights_host(tf, e->N, d, w);
        HIPCHK(hipMemcpyAsync(e->d_target, tf.data(), nf * sizeof(float), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->d_lossw, w.data(), (size_t)d * sizeof(float), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        e->have_target = true;
        return ISK_OK;
    }
    IsaArgs ia;
    ia.ks = e->d_ks; ia.ks64 = e->d_ks64; ia.chi = e->d_out + e->N * e->K * d; ia.work = e->d_isa_work; ia.target = e->d_target;
    ia.lossw = e->d_lossw; ia.status = e->d_status; ia.N = e->N; ia.d = d;
    ia.permute = e->cfg.isa_permute; ia.whitening = e->cfg.isa_whitening;
    ia.out_first = 0; ia.out_count = e->N;
    isk_launch_isa(ia, e->stream);
    e->have_target = true;
    if (check) CHK(check_status(e));
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
int pick_slabs(int B) {
    const int nchunks = (B + 15) / 16;
    int s = (nchunks + 47) / 48;                  // about three 16-row chunks per wave
    return s < 1 ? 1 : (s > ISK_MAX_SLABS ? ISK_MAX_SLABS : s);
}
// bring the device copy of [beta1^t, beta2^t] up to date (after create / set_opt_state / reset, or steps taken in host mode);
// synchronous and rare.  Never called while the stream is being captured.
bool prepare_dev_bt(isk_engine* e) {
    if (e->bt_dev_valid) return true;
    const double h[2] = {e->bt1, e->bt2};
    if (hipMemcpyAsync(e->d_bt, h, sizeof(h), hipMemcpyHostToDevice, e->stream) != hipSuccess) return false;
    if (hipStreamSynchronize(e->stream) != hipSuccess) return false;
    e->bt_dev_valid = true;
    return true;
}

// row slabs + the train_grad1n variant: while the throttled featurizer shares the CUs the 8-wave variant (80 KB LDS instead of
// 158 KB) lets both fit; option grad1_nw8 pins one variant (tests: the benchmarked instance without the side stream)
static int grad1_flags(const isk_engine* e, int nslab) {
    const bool nw8 = e->opt.grad1_nw8 >= 0 ? e->opt.grad1_nw8 != 0 : e->ys_ahead;
    return nslab | (nw8 ? 0x100 : 0);
}

// one optimizer step's gradient into d_grad (slab 0 .. nslab-1).  step: an optimizer step follows (launch_opt), so
// train_tail hands the Adam bias corrections of this step over on the device
// the arguments of this step's optimizer kernel (opt_step, or train_grad1n carrying it)
static OptArgs opt_args(isk_engine* e, float gscale, bool publish) {
    OptArgs a;
    a.theta = e->d_theta; a.grad = e->d_grad; a.m = e->d_m; a.v = e->d_v; a.n = e->nparams;
    a.kind = e->cfg.opt_kind; a.lr = e->cfg.lr; a.reg = e->cfg.reg; a.beta1 = e->cfg.beta1; a.beta2 = e->cfg.beta2;
    a.eps = e->cfg.opt_eps; a.rho = e->cfg.rho;
    a.bc1 = (float)(1.0 - e->bt1); a.bc2 = (float)(1.0 - e->bt2);
    // fused path: the bias corrections of this step were handed over on the device by train_tail (launch_grads)
    const bool dev = !e->layered && e->bt_dev_valid;
    a.bt = dev ? e->d_bt + 2 : nullptr;
    a.gscale = gscale; a.loss_acc = e->d_loss_acc;
    a.host_pub = publish ? e->d_hres : nullptr;
    a.nslab = e->last_nslab;
    a.slab_prefix = e->off_b[0] + e->cfg.widths[1];          // [LN | W1 | b1]: the blocks train_grad1n writes per row slab
    return a;
}

// fuse_mode: 0 = the gradient goes to d_grad; 1 = the optimizer step rides on train_grad1n (the caller then runs launch_opt with
// kernel_done = true for the host-side bookkeeping); 2 = the same, and it is the epoch's last step (publishes the result)
bool can_fuse_opt(const isk_engine* e, int B, int nslab) {
    if (!e->opt.fuse_opt || e->layered || !e->native) return false;
    const bool nw8 = e->opt.grad1_nw8 >= 0 ? e->opt.grad1_nw8 != 0 : e->ys_ahead;
    return isk_train_grad1n_can_fuse(train_shape(e), e->Fp, B, nslab, nw8 ? 1 : 0);
}

// row slabs of a single-GPU training step: batches of up to fuse_opt_rows rows stay on ONE slab where that lets the optimizer
// step ride on train_grad1n (a longer MFMA chain per wave, but no opt_step launch and no slab sum)
int pick_slabs_fused(const isk_engine* e, int B) {
    if (B <= e->opt.fuse_opt_rows && can_fuse_opt(e, B, 1)) return 1;
    return pick_slabs(B);
}

void launch_grads(isk_engine* e, const int* d_idx, int B, float inv_div, int nslab, bool step, bool capturing, int fuse_mode) {
    if (e->layered) { launch_grads_layered(e, d_idx, B, inv_div); return; }
    const TrainShape s = train_shape(e);
    TrainBuffers b = train_buffers(e);
    int cpb;
    isk_fwd1n_plan(B, e->Fp, isk_pad16(e->cfg.widths[1]) / 16, &b.KS, &cpb);
    b.zstride = z1_rows(B) * isk_pad16(e->cfg.widths[1]);
    if (step && (capturing ? e->bt_dev_valid : prepare_dev_bt(e))) b.bt = e->d_bt;
    if (e->opt.fuse_tail && e->opt.tail_narrow && e->d_tail_cnt && isk_train_fwdtail_ok(s)) {
        isk_launch_train_fwdtail(s, b, d_idx, B, inv_div, e->d_tail_cnt, e->stream);
        e->last_fwdtail = true;
    } else {
        e->last_fwdtail = false;
        isk_launch_train_fwd1n(s, b, d_idx, B, e->stream);
        isk_launch_train_tail(s, b, d_idx, B, inv_div, e->stream);
    }
    e->last_nslab = nslab;
    if (fuse_mode) {
        const OptArgs oa = opt_args(e, 1.0f, fuse_mode == 2);
        isk_launch_train_grad1n(s, b, d_idx, B, grad1_flags(e, nslab), e->stream, &oa);
    } else {
        isk_launch_train_grad1n(s, b, d_idx, B, grad1_flags(e, nslab), e->stream);
    }
}

// fold: refresh the weight images the forward / training kernels read.
// publish: this is the epoch's LAST step -- it writes the result into mapped pinned host memory (a kernel ends only when its
// stores are acknowledged: publishing from every step cost ~2 us per step, 0.2 ms per iteration at minibatch = 100)
// layered path without LayerNorm: the MFMA training GEMMs read theta (or their own images of it), so the forward weight
// images are folded once, behind the epoch's last step (every epoch driver asks here: the per-step fold is ~38 us at C5)
bool epoch_lazy_fold(isk_engine* e) { return e->layered && !e->cfg.layernorm && (bf16_train_active(e) || mfma32_train_active(e)); }
// ... and not at all where nothing reads them: the fused wide forward streams its own fragment-linear images of theta and the bf16
// GEMMs theirs (a forward of foreign rows through the generic layered kernels folds on demand: wt_dirty stays set)
bool epoch_no_fold(isk_engine* e) { return epoch_lazy_fold(e) && e->xh16 && (bf16_train_active(e) || mfma32_train_active(e)); }

void launch_opt(isk_engine* e, float gscale, bool fold, bool capturing, bool publish, const OptPeers* peers, bool kernel_done) {
    const bool dev = !e->layered && e->bt_dev_valid;
    (void)capturing;
    bool images_fresh = false;                   // the wide-network path wrote the next step's weight images with the step
    if (!kernel_done) {                          // (kernel_done: train_grad1n carried the step -- launch_grads with fuse_mode)
        const OptArgs a = opt_args(e, gscale, publish);
        if (peers) isk_launch_opt_step_peers(a, *peers, e->stream);
        else if (layered_opt_boundary(e, a)) images_fresh = true;
        else isk_launch_opt_step(a, e->stream);
    }
    e->next_idx = nullptr;
    e->loss_acc_zeroed = false;
    e->bt1 *= e->cfg.beta1; e->bt2 *= e->cfg.beta2;
    if (!dev) e->bt_dev_valid = false;           // this step advanced only the host copy
    e->b16.wt_dirty = !images_fresh; e->b16.wf_dirty = true;
    if (e->native) { e->wt_dirty = true; e->w16_dirty = true; }      // the training steps read theta; forwards re-fold on demand
    else if (fold) launch_fold(e);
    else e->wt_dirty = true;                     // a forward folds on demand (ensure_folded)
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
int stage_perm_ahead(isk_engine* e, const int32_t* perm) {
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

// The whole epoch as ONE persistent launch (kernels_train_epoch.hip): fused-path networks with a narrow tail on a single GPU.
// false: not applicable here (or a launch failed once): the caller runs the launch chain.
static bool train_epoch_persist(isk_engine* e, long long bs, long long nb) {
    e->last_epoch_persist = false;
    if (!e->opt.train_persist || e->tp_disabled || e->layered || !e->native || nb < 1 || e->ncu < 1) return false;
    const TrainShape s = train_shape(e);
    EpochPlan p;
    const bool chain = e->opt.train_persist == 2;              // small batches on the workgroups of one XCD (kernels_train_chain.hip)
    if (!(chain ? isk_train_chain_plan(s, e->Fp, (int)bs, &p) : isk_train_epoch_plan(s, e->Fp, (int)bs, e->ncu, &p))) return false;
    const int H1p = isk_pad16(e->cfg.widths[1]);
    const size_t zp_floats = (size_t)p.KS * p.zrows * H1p, lp_floats = (size_t)p.KS * p.nt * 2 * p.cps * 16;
    if (e->ep_zp_floats < zp_floats) {
        (void)hipStreamSynchronize(e->stream);
        free_dev(e->d_ep_zp); e->d_ep_zp = nullptr; e->ep_zp_floats = 0;
        if (hipMalloc((void**)&e->d_ep_zp, zp_floats * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); e->tp_disabled = true; return false; }
        e->ep_zp_floats = zp_floats;
    }
    if (e->ep_lnpart_floats < lp_floats) {
        (void)hipStreamSynchronize(e->stream);
        free_dev(e->d_ep_lnpart); e->d_ep_lnpart = nullptr; e->ep_lnpart_floats = 0;
        if (hipMalloc((void**)&e->d_ep_lnpart, lp_floats * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); e->tp_disabled = true; return false; }
        e->ep_lnpart_floats = lp_floats;
    }
    if (!e->d_ep_cnt && hipMalloc((void**)&e->d_ep_cnt, ISK_EP_CNT_WORDS * sizeof(unsigned)) != hipSuccess) {
        (void)hipGetLastError(); e->tp_disabled = true; return false;
    }
    EpochArgs a;
    std::memset(&a, 0, sizeof(a));
    a.X = e->d_xs32; a.Fp = e->Fp; a.perm = e->d_perm; a.B = (int)bs; a.nb = (int)nb;
    a.theta = e->d_theta; a.m = e->d_m; a.v = e->d_v;
    a.off_lns = e->off_lns; a.off_lnb = e->off_lnb; a.off_W0 = e->off_W[0]; a.off_b0 = e->off_b[0];
    a.ln = e->cfg.layernorm; a.nch = e->Fp / 16;
    a.target = e->d_target; a.lossw = e->d_lossw; a.inv_div = 1.0f / (float)bs;
    a.zp = e->d_ep_zp; a.delta1 = e->d_delta1; a.part = e->d_part; a.lnpart = e->d_ep_lnpart; a.cnt = e->d_ep_cnt;
    a.opt_kind = e->cfg.opt_kind; a.lr = e->cfg.lr; a.reg = e->cfg.reg; a.beta1 = e->cfg.beta1; a.beta2 = e->cfg.beta2;
    a.eps = e->cfg.opt_eps; a.rho = e->cfg.rho;
    a.bt1 = e->bt1; a.bt2 = e->bt2;
    a.loss_acc = e->d_loss_acc; a.host_pub = e->d_hres;
    const bool stamps = e->opt.debug_stamps != 0;
    if (stamps) {
        if (!e->dbg_st) (void)hipMalloc((void**)&e->dbg_st, (size_t)4096 * 12 * 8);
        (void)hipMemsetAsync(e->dbg_st, 0, 64 * 8, e->stream);
        a.dbg = e->dbg_st;
    }
    if ((chain ? isk_launch_train_chain(s, p, a, e->stream) : isk_launch_train_epoch(s, p, a, e->stream)) != 0) { (void)hipGetLastError(); e->tp_disabled = true; return false; }
    // what launch_grads / launch_opt keep track of on the host
    for (long long b = 0; b < nb; ++b) { e->bt1 *= e->cfg.beta1; e->bt2 *= e->cfg.beta2; }
    e->bt_dev_valid = false;             // the device copy of beta^t did not take part
    e->last_nslab = 1;
    e->w16_dirty = true; e->wt_dirty = true;
    e->b16.wt_dirty = e->b16.wf_dirty = true;
    e->loss_acc_zeroed = false;
    e->last_epoch_persist = true;
    if (stamps) {
        (void)hipStreamSynchronize(e->stream);
        unsigned long long h[64];
        (void)hipMemcpy(h, e->dbg_st, sizeof(h), hipMemcpyDeviceToHost);
        static const char* nm_ep[14] = {"step", "P1", "pubA", "waitA+C", "Kreduce", "tail", "pubB", "waitB", "P3", "loss+E1", "E2+pubSib", "E4", "E5", "E6+pubC"};
        static const char* nm_ch[14] = {"step", "A", "pubA", "waitA", "Kreduce", "tail", "pubB+xload", "waitB", "stage-delta1", "G-mfma", "D+prefetch", "opt", "-", "-"};
        const char* const* nm = chain ? nm_ch : nm_ep;
        for (int g = 0; g < 2; ++g) {
            fprintf(stderr, "[epoch stamps] wg %s:", g == 0 ? "0 (tail role)" : "G-1");
            unsigned long long prev = h[32 * g];
            for (int k = 1; k < 14; ++k) {
                if (!h[32 * g + k]) continue;
                fprintf(stderr, " %s %.2f", nm[k], (double)(h[32 * g + k] - prev) * 0.01);
                prev = h[32 * g + k];
            }
            fprintf(stderr, " | total %.2f us\n", (double)(prev - h[32 * g]) * 0.01);
        }
    }
    return true;
}

// The epoch's nb x 5 dependent launches as ONE hipGraph launch.  A dependent kernel launched from the host costs ~3.0 us
// on this stack, the same chain replayed from a graph ~1.9 us (measured with a 50-kernel chain of empty kernels); the
// training chain is nothing but such launches (50 per iteration at C3, 500 at the reference's default minibatch of 100).
// Every launch argument is constant from one epoch to the next: the batch slices are fixed offsets into d_perm (its
// CONTENT is what the shuffle changes), and the Adam bias corrections are read from d_bt on the device.  The graph is
// captured from the ordinary launch functions the first time and re-captured when the batch plan, the train_grad1
// variant or the buffers change.  ISK_TRAIN_GRAPH=0 (or any capture error) keeps the per-kernel launches.
static bool train_epoch_graph(isk_engine* e, long long bs, long long nb) {
    if (!e->opt.train_graph || e->tg_disabled || e->layered || nb < 2) return false;
    const int nslab = pick_slabs_fused(e, (int)bs);
    const int flags = grad1_flags(e, nslab);
    if (!prepare_dev_bt(e)) return false;
    const float* xs_rows = e->d_xs32;                    // part of the key: the buffer pair swaps every iteration
    isk_engine::TrainGraph* tg = nullptr;
    for (auto& g : e->tg)
        if (g.exec && g.xs == xs_rows) tg = &g;
    if (!tg) {                                           // a free slot, else the one that does not belong to the other live buffer
        tg = !e->tg[0].exec ? &e->tg[0] : (!e->tg[1].exec ? &e->tg[1] : (e->tg[0].xs == e->d_xs32_alt ? &e->tg[1] : &e->tg[0]));
    }
    const int flags_f = flags | (can_fuse_opt(e, (int)bs, nslab) ? 0x1000 : 0) | (e->opt.fuse_tail ? 0x2000 : 0) | (e->opt.tail_narrow ? 0x4000 : 0);        // (the fused step is part of what the graph was captured for)
    if (!tg->exec || tg->bs != bs || tg->nb != nb || tg->flags != flags_f || tg->xs != xs_rows) {
        if (tg->exec) { (void)hipGraphExecDestroy(tg->exec); tg->exec = nullptr; }
        const double s1 = e->bt1, s2 = e->bt2;
        if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            e->tg_disabled = true;
            return false;
        }
        const bool fuse = can_fuse_opt(e, (int)bs, nslab);
        for (long long b = 0; b < nb; ++b) {
            launch_grads(e, e->d_perm + b * bs, (int)bs, 1.0f / (float)bs, nslab, true, true, fuse ? (b == nb - 1 ? 2 : 1) : 0);
            launch_opt(e, 1.0f, true, true, b == nb - 1, nullptr, fuse);
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
        tg->bs = bs; tg->nb = nb; tg->flags = flags_f; tg->xs = xs_rows;
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
// the host side of an epoch's end: wait for the event behind its last optimizer step, then read what that step published
static int finish_epoch(isk_engine* e, double* loss, int* status_word) {
    HIPCHK(hipEventSynchronize(e->ev_loss));                     // the shuffle order's copy (pinned staging) precedes it on the stream
    HIPCHK(hipGetLastError());
    const double* acc = e->h_res;
    if (status_word) std::memcpy(status_word, acc + 2, sizeof(int));
    if (acc[1] == 2.0) {                                         // the persistent epoch kernel gave up on a hand-off (bounded spin)
        e->tp_disabled = true;
        return fail(ISK_EHIP, "the persistent training-epoch kernel timed out on an in-launch hand-off; parameters are undefined (option train_persist = 0 avoids it)");
    }
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
    if (!e->loss_acc_zeroed) HIPCHK(zero_loss_acc(e));
    e->loss_acc_zeroed = false;
    {
        ScopedTimer t(e, ISK_T_TRAIN);
        const bool lazy_fold = epoch_lazy_fold(e);
        if (!train_epoch_persist(e, bs, nb) && !train_epoch_graph(e, bs, nb)) {
            const int nsl = pick_slabs_fused(e, (int)bs);
            const bool fuse = can_fuse_opt(e, (int)bs, nsl);
            EpochGatherScope images(e, e->d_perm, bs, nb);
            for (long long b = 0; b < nb; ++b) {
                launch_grads(e, e->d_perm + b * bs, (int)bs, 1.0f / (float)bs, nsl, true, false, fuse ? (b == nb - 1 ? 2 : 1) : 0);
                e->next_idx = b + 1 < nb ? e->d_perm + (b + 1) * bs : nullptr; e->next_B = (int)bs;
                launch_opt(e, 1.0f, !lazy_fold || (b == nb - 1 && !epoch_no_fold(e)), false, b == nb - 1, nullptr, fuse);
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
            const int early = e->opt.side_early;
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
