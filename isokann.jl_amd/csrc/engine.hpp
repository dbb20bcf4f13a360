// Shared declarations of the host engine's translation units (engine*.hip): the engine state, the error macros and the
// helpers one part calls in another.  The C ABI itself is include/isokann_hip.h; the kernel launchers are isk_internal.h.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>
#include <condition_variable>
#include <mutex>

#include <rccl/rccl.h>
#include "isk_internal.h"

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
int isk_fail(int code, const std::string& msg);        // sets the thread-local message of isk_last_error, returns code
#define fail isk_fail

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

// ---------------------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------------------
struct EventPair { hipEvent_t a, b; int id; };

// the engines one host process drives together (isk_dp_create_all): what the option dp_peer_reduce needs beyond the communicator
struct isk_engine;
struct DpGroup {
    std::vector<isk_engine*> eng;
    // a reusable host barrier for the per-engine driver threads of isk_dp_run_all (an event must have been RECORDED by its
    // owner's thread before another thread may make its stream wait for it)
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long long generation = 0;
    bool broken = false;               // a member left the run early: nobody waits any more
    bool wait() {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        const long long gen = generation;
        if (++arrived == (int)eng.size()) { arrived = 0; ++generation; cv.notify_all(); return true; }
        cv.wait(lk, [&] { return generation != gen || broken; });
        return !broken;
    }
    void leave() { std::lock_guard<std::mutex> lk(mu); broken = true; cv.notify_all(); }
    void reset() { std::lock_guard<std::mutex> lk(mu); broken = false; arrived = 0; }
};

struct isk_engine {
    isk_config cfg;
    IskOptions opt;                  // per-engine knobs (isk_set_option / isk_get_option)
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
    bool xh16 = false;               // layered + plain-bf16 forward + fused wide kernel: d_xhat holds bf16 rows (decided at isk_set_data)
    void* d_W16L[ISK_MAX_LAYERS] = {nullptr};     // bf16 planes per layer (forward_mode 1)
    float* d_WtT[ISK_MAX_LAYERS] = {nullptr};     // fragment-linear W' per layer l >= 1 (delta back-propagation)
    float *d_actA = nullptr, *d_actB = nullptr;   // resident forward ping-pong [(N*K+N)][max pad32(H)]
    float* d_lz[ISK_MAX_LAYERS] = {nullptr};      // training: pre-activations per layer [Bpad][pad16(N_l)]
    float* d_la[ISK_MAX_LAYERS] = {nullptr};      //           activations
    float* d_ld[ISK_MAX_LAYERS] = {nullptr};      //           deltas
    float *d_G0 = nullptr, *d_D0 = nullptr;       // raw layer-0 gradient for the LayerNorm unfold
    // bf16 training GEMMs (option train_bf16; kernels_gemm16.hip): images of the batch, of the activations / deltas (row-major and
    // transposed) and of the weights, allocated on first use for the resident N
    struct Bf16Train {
        long long rows = 0;                           // Bp the buffers were sized for (0: not allocated)
        int esz = 0;                                  // bytes per image element they were sized for: 2 (bf16) or 4 (Float32 arithmetic)
        int wt_esz = 0;                               // element size of what the Wt images currently hold
        void *xb = nullptr, *xt = nullptr;            // [Bp][Fp], [F + 1][Bp]
        void* a16[ISK_MAX_LAYERS] = {nullptr};        // act_l          [Bp][pad32(N_l)]
        void* at16[ISK_MAX_LAYERS] = {nullptr};       // act_l'         [N_l + 1][Bp], row N_l = ones
        void* d16[ISK_MAX_LAYERS] = {nullptr};        // delta_l        [Bp][pad32(N_l)]
        void* dt16[ISK_MAX_LAYERS] = {nullptr};       // delta_l'       [N_l][Bp]
        void* Wt[ISK_MAX_LAYERS] = {nullptr};         // W_l' image     [N_l][pad32(K_l)]      (training GEMMs)
        void* Wf[ISK_MAX_LAYERS] = {nullptr};         // the same numbers, fragment-linear [pad32(K_l) / 32][32][64][8] (fused wide forward; hidden layers <= 512 wide)
        float* G0 = nullptr;                          // LayerNorm: raw layer-0 gradient [F + 1][N_0] (row F = D)
        float* head_part = nullptr;                   // head kernel: per-workgroup partials [ISK_HEAD_MAXG][head_stride] of dW_L | db_L | loss
        long long head_stride = 0;
        bool wt_dirty = true, wf_dirty = true;
        // the batch images of a whole epoch, gathered in ONE launch in front of its steps (layered_epoch_gather)
        void *xb_all = nullptr, *xt_all = nullptr;    // [nb][Bp][Fp], [nb][F + 1][Bp]
        size_t xb_all_bytes = 0, xt_all_bytes = 0;
        const int* pre_idx = nullptr;                 // the batch whose images the last step-boundary launch wrote into xb / xt (or nullptr)
        int pre_B = 0, pre_esz = 0;
        const int* ep_perm = nullptr;                 // valid while ep_nb > 0: the epoch's shuffle order, batch size, batch count
        long long ep_bs = 0, ep_nb = 0;
        int ep_esz = 0;
    } b16;
    float* d_zero = nullptr;                      // zero bias for the transposed GEMMs
    double* d_loss_acc = nullptr;
    int* d_status = nullptr;
    int* d_rank_err = nullptr;       // data-parallel runs: this rank's error flag (set by the host on a local failure), reduced with the gradients
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
    // the persistent epoch kernel (kernels_train_epoch.hip): hand-off buffers, sized for the batch plan they were allocated for
    int ncu = 0;                         // compute units of the device (its grid must be co-resident)
    float* d_ep_zp = nullptr; size_t ep_zp_floats = 0;
    float* d_ep_lnpart = nullptr; size_t ep_lnpart_floats = 0;
    unsigned* d_ep_cnt = nullptr;
    // bucketed gradient all-reduce of the layered path (engine_dp_native.hip): the blocks of layers >= 1 are reduced on a stream of
    // their own while the first layer's backward products run
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_b1_ready = nullptr, ev_b1_done = nullptr;
    void (*grad_hook)(isk_engine*) = nullptr;   // called by the layered backward once every block of layers >= 1 is enqueued
    int bucket_issued = 0;                      // the hook ran in this step: the upper bucket is already on its way
    int bucket_err = 0;                         // ncclResult of the hook's collective (0 = success)
    bool tp_disabled = false;
    const int* next_idx = nullptr;              // set by the epoch loops around launch_opt: the rows of the NEXT optimizer step (or nullptr)
    int next_B = 0;
    bool last_fwdtail = false;                  // the last launch_grads took the one-launch forward + tail (option fuse_tail)
    bool last_epoch_persist = false;            // a launch failed or timed out: the four-kernel chain from then on
    unsigned long long* dbg_st = nullptr;            // diagnostics of forward_resident (options debug_stamps / debug_rowmod)
    int* dbg_rows = nullptr;
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
    unsigned* d_tail_cnt = nullptr;          // arrival counters of train_fwdtail_kernel (one line per 32-row block, zero between launches)
    int* d_perm = nullptr;
    hipEvent_t ev_perm = nullptr;   // the shuffle order of the coming epoch was uploaded on the side stream (isk_run)
    double* h_res = nullptr;     // pinned + mapped: [loss sum, collapsed flag, status word], published by opt_step (d_hres = its device address)
    double* d_hres = nullptr;
    bool loss_acc_zeroed = false; // the K-mean kernel of this iteration already zeroed d_loss_acc
    int* h_perm = nullptr;       // pinned staging for the shuffle order: an async copy from pageable memory waits for the stream
    double* d_At = nullptr;      // host-driven data-parallel ISA: the Float64 simplex inverse (applied in Float64, as on the device path)
    int* d_rowperm = nullptr;
    double* d_isa_work = nullptr;
    double* d_ks64 = nullptr;                     // K chi of the resident data as the K-mean summed it in double (IsaArgs.ks64)
    double* d_gks64 = nullptr;                    // ... all-gathered (native data-parallel run)
    // native data-parallel run (isk_dp_run): RCCL communicator + gathered ks / chi
    ncclComm_t comm = nullptr;
    int dp_rank = 0, dp_world = 1;
    DpGroup* grp = nullptr;              // set by isk_dp_create_all (shared by the group; engine 0 owns it)
    bool last_run_peer = false;          // the last isk_dp_run summed the replicas' gradients over peer access (no all-reduce)
    float* d_grad_base = nullptr;        // d_grad alternates between two slabs of its allocation in the peer-reduce run
    hipEvent_t ev_grads[2] = {nullptr, nullptr};   // recorded behind the gradient kernels of an optimizer step (by step parity)
    int dp_err = ISK_OK;             // native run: this rank's first failure inside the step loop (dp_poison) ...
    std::string dp_errmsg;           // ... and its message; reported when the epoch's collectives are through
    int* h_flag = nullptr;           // pinned staging word for d_rank_err
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

// ---- timers ---------------------------------------------------------------------------
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

struct TmpBuf {
    void* p = nullptr;
    ~TmpBuf() { if (p) (void)hipFree(p); }
};

// ---- engine.hip ------------------------------------------------------------------------
int layer_kp(const isk_engine* e, int l);
void free_dev(void* p);
void shuffle_perm(isk_engine* e, std::vector<int32_t>& p);
int act_of_layer(const isk_engine* e, int l);
MlpFwdArgs fwd_args(const isk_engine* e, const float* X, long long S, float* out);
TrainShape train_shape(const isk_engine* e);
long long z1_rows(long long B);
TrainBuffers train_buffers(const isk_engine* e);
void launch_fold(isk_engine* e);
void ensure_folded(isk_engine* e);
void fold_ahead(isk_engine* e);
void api_to_engine(const isk_engine* e, const float* a, std::vector<float>& out);
void engine_to_api(const isk_engine* e, const float* in, float* a);
FeatArgs feat_args(const isk_engine* e);                 // featurizer arguments common to every launch (pairs, shape, LayerNorm mode, options)
void featurize_resident(isk_engine* e);
void featurize_ys_beside_training(isk_engine* e, long long train_launch_steps);
void featurize_early_share(isk_engine* e, long long train_launch_steps);
void side_featurize_hook(isk_engine* e);
int launch_forward(isk_engine* e, const MlpFwdArgs& f);
int need_data(isk_engine* e);
int forward_resident(isk_engine* e, int which, bool ahead = false);
void launch_kmean(isk_engine* e);
hipError_t zero_loss_acc(isk_engine* e);
int check_status(isk_engine* e);
int compute_target(isk_engine* e, bool check, bool ahead = false);
int dp_isa_dev(isk_engine* e, const float* ks_all_dev, const double* ks64_all_dev, const float* chi_all_dev, long long n_global, long long first);
int pick_slabs(int B);
void layered_epoch_gather(isk_engine* e, const int* d_perm, long long bs, long long nb);
void layered_epoch_end(isk_engine* e);
struct OptArgs;
bool layered_opt_boundary(isk_engine* e, const OptArgs& a);
struct EpochGatherScope {            // the epoch images are valid only while the epoch's steps are being enqueued
    isk_engine* e;
    EpochGatherScope(isk_engine* e_, const int* d_perm, long long bs, long long nb) : e(e_) { layered_epoch_gather(e, d_perm, bs, nb); }
    ~EpochGatherScope() { layered_epoch_end(e); }
};
int pick_slabs_fused(const isk_engine* e, int B);
bool prepare_dev_bt(isk_engine* e);
void launch_grads(isk_engine* e, const int* d_idx, int B, float inv_div, int nslab = 1, bool step = true, bool capturing = false, int fuse_mode = 0);
bool can_fuse_opt(const isk_engine* e, int B, int nslab);
void launch_opt(isk_engine* e, float gscale, bool fold = true, bool capturing = false, bool publish = true, const OptPeers* peers = nullptr,
                bool kernel_done = false);
bool epoch_lazy_fold(isk_engine* e);             // the forward weight images may wait for the epoch's last optimizer step
bool epoch_no_fold(isk_engine* e);               // ... or for a forward that actually reads them
int stage_perm_ahead(isk_engine* e, const int32_t* perm);
void engine_options_init(isk_engine* e);                 // defaults, then ISK_<KEY> of the creating process
// ---- engine_layered.hip ----------------------------------------------------------------
int forward_layered(isk_engine* e, const float* X, long long S, float* out, float* actA, float* actB);
void launch_grads_layered(isk_engine* e, const int* d_idx, int B, float inv_div);
void bf16_train_release(isk_engine* e);
bool bf16_train_active(const isk_engine* e);
bool mfma32_train_active(const isk_engine* e);   // layered f32 training on the in-house NT GEMMs in Float32 arithmetic
int bf16_weight_images_alloc(isk_engine* e);             // isk_create, layered engines: the [max(N, 512)][pad32 K] bf16 images W_l'
void bf16_weight_images_ensure(isk_engine* e, bool want_wt, bool want_wf);           // refresh them if theta changed since they were built
int bf16_planes(const isk_engine* e);
bool wide_forward_shape_ok(const isk_engine* e);         // the fused wide forward covers this network (hidden widths <= 512, nout <= 16)
// ---- engine_isa_host.hip ---------------------------------------------------------------
// ISA (src/isotarget.jl:74-127): up to ISK_ISA_DEV_MAXD outputs the whole target runs on the device (kernels_isa.hip: fixperm as the
// reference's search over d! permutations); beyond, chi and K chi come to the host for the Float64 solve with fixperm as a linear
// assignment problem (engine_isa_host.hip) -- as the reference itself does its ISA on the host (`|> cpu`, src/isotarget.jl:101-103)
#define ISK_ISA_DEV_MAXD 8
#define ISK_ISA_HOST_MAXD 64
std::vector<int> assign_min_cost(const std::vector<double>& Cst, int d);
int isa_solve_host(const isk_engine* e, const std::vector<float>& ks, const std::vector<float>& chi, long long N, int d,
                   std::vector<double>& At, std::vector<int>& rowperm, std::vector<double>* target_out);
void loss_weights_host(const std::vector<float>& tgt, long long N, int d, std::vector<float>& w);
// ---- engine_dp.hip ---------------------------------------------------------------------
void dp_comm_release(isk_engine* e);
