// Native data-parallel run over N: RCCL (over xGMI) called by the engine on its own stream, so an iteration has no host code
// between the kernels of a step.  Two host models share this file:
//   * one process per GPU (torch.distributed.run, MPI): isk_dp_comm_id / isk_dp_comm_init / isk_dp_run
//   * ONE process for all GPUs of a node, as the reference's single Julia process (src/iso.jl:256, SURVEY.md 8b "Threading"):
//     isk_dp_create_all (ncclCommInitAll) / isk_dp_set_data_all / isk_dp_run_all -- every engine is driven by a host thread
//     of its own inside the one blocking call, because the step loop is launch-bound: 8 engines x ~40 launches per epoch
//     from a single thread would serialise ~1 ms of launch overhead per iteration.
// Error agreement: argument and allocation failures are agreed on BEFORE the first collective of a call (one small all-reduce),
// and every gradient all-reduce carries a flag slot behind the batch loss, so a rank-local failure inside the loop skips the
// update on every rank in the same step and every rank returns an error (its own, or ISK_EPEER) instead of waiting for a peer.
// The reference has no multi-device code at all.
#include <dlfcn.h>

#include <thread>

#include "engine.hpp"

// ---------------------------------------------------------------------------------------
// RCCL, loaded lazily so the library has no link-time dependency on it
// ---------------------------------------------------------------------------------------
namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;     // optional: single-process multi-device
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
    // ISK_RCCL_LIB: a process-wide loader path (another collective library with the same entry points: tests/mock_rccl), not an
    // engine option -- a library is loaded once per process.  Otherwise the copy the host process already uses first (PyTorch
    // ships its own librccl: two different RCCL builds in one process would each keep their own view of the devices)
    void* h = nullptr;
    if (const char* over = getenv("ISK_RCCL_LIB")) {
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
    a.CommInitAll = (decltype(a.CommInitAll))dlsym(h, "ncclCommInitAll");
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

void dp_comm_release(isk_engine* e) {
    if (e->comm && g_rccl.CommDestroy) { (void)hipStreamSynchronize(e->stream); (void)g_rccl.CommDestroy(e->comm); }
    e->comm = nullptr;
    e->dp_rank = 0; e->dp_world = 1;
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

// collectives run when there is more than one rank (option dp_force_coll: also on a single-rank communicator, for tests and
// for rehearsing one GPU's share of a multi-GPU run)
static bool dp_collectives(const isk_engine* e) { return e->comm != nullptr && (e->dp_world > 1 || e->opt.dp_force_coll); }

// what the coming isk_dp_run needs beside the resident data: allocated BEFORE the first collective of the call, so that an
// allocation failure is part of the up-front agreement instead of a rank that never arrives at an all-gather
static int dp_prepare(isk_engine* e) {
    if (!dp_collectives(e) || e->cfg.target_kind != ISK_TARGET_ISA) return ISK_OK;
    const int d = e->nout;
    const long long n_global = e->N * e->dp_world;
    if (e->gather_rows < n_global) {
        HIPCHK(hipStreamSynchronize(e->stream));
        free_dev(e->d_gks); free_dev(e->d_gchi); free_dev(e->d_gks64);
        e->d_gks = e->d_gchi = nullptr; e->d_gks64 = nullptr; e->gather_rows = 0;
        HIPCHK(hipMalloc((void**)&e->d_gks, (size_t)n_global * d * sizeof(float)));
        HIPCHK(hipMalloc((void**)&e->d_gchi, (size_t)n_global * d * sizeof(float)));
        HIPCHK(hipMalloc((void**)&e->d_gks64, (size_t)n_global * d * sizeof(double)));
        e->gather_rows = n_global;
    }
    if (e->isa_work_dp_rows < n_global) {
        HIPCHK(hipStreamSynchronize(e->stream));
        free_dev(e->d_isa_work_dp);
        e->d_isa_work_dp = nullptr; e->isa_work_dp_rows = 0;
        HIPCHK(hipMalloc((void**)&e->d_isa_work_dp, (size_t)isk_isa_work_doubles(n_global, d) * sizeof(double)));
        e->isa_work_dp_rows = n_global;
    }
    return ISK_OK;
}

// One all-reduce(max) of [N, -N, local status] before anything else of the call: every rank learns whether the shards are equal
// (the gathered ISA arrays and the step count assume it) and whether some rank failed its argument checks or allocations.
// Returns the rank's own error if it has one, ISK_EPEER if only a peer has.
static int dp_agree(isk_engine* e, int local_status, const std::string& local_msg) {
    long long* d_n = nullptr;
    hipError_t he = hipMalloc((void**)&d_n, 3 * sizeof(long long));
    long long hn[3] = {e->N, -e->N, (long long)(he != hipSuccess ? ISK_ENOMEM : local_status)};
    long long gn[3] = {0, 0, 0};
    ncclResult_t r = ncclSuccess;
    if (he == hipSuccess) {
        he = hipMemcpyAsync(d_n, hn, sizeof(hn), hipMemcpyHostToDevice, e->stream);
        if (he == hipSuccess) r = g_rccl.AllReduce(d_n, d_n, 3, ncclInt64, ncclMax, e->comm, e->stream);
        if (he == hipSuccess && r == ncclSuccess) he = hipMemcpyAsync(gn, d_n, sizeof(gn), hipMemcpyDeviceToHost, e->stream);
        const hipError_t se = hipStreamSynchronize(e->stream);
        if (he == hipSuccess) he = se;
        (void)hipFree(d_n);
    }
    if (local_status != ISK_OK) return fail(local_status, local_msg);
    if (r != ncclSuccess) return fail(ISK_EHIP, std::string("rccl: ") + g_rccl.GetErrorString(r));
    if (he != hipSuccess) return fail(ISK_EHIP, std::string("data-parallel agreement: ") + hipGetErrorString(he));
    if (gn[2] != ISK_OK)
        return fail(ISK_EPEER, std::string("a peer rank reported an error before the run started (") + isk_status_string((int)gn[2]) + ")");
    if (gn[0] != e->N || gn[1] != -e->N) return fail(ISK_EINVAL, "isk_dp_run needs equal shards on every rank");
    return ISK_OK;
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
    if (d > ISK_ISA_DEV_MAXD && !coll) return compute_target(e, false, e->cfg.featurize_every_iteration != 0);   // one rank: the host solve of isk_run
    CHK(forward_resident(e, 3, e->cfg.featurize_every_iteration != 0));
    side_featurize_hook(e);
    ScopedTimer t(e, ISK_T_TARGET);
    launch_kmean(e);
    const float* ks = e->d_ks;
    const double* ks64 = e->d_ks64;                       // K chi unrounded (the K-mean's double sums): what the Float64 solve reads
    const float* chi = e->d_out + e->N * e->K * d;
    long long n_global = e->N, first = 0;
    if (coll) {
        n_global = e->N * e->dp_world; first = e->N * e->dp_rank;
        if (e->gather_rows < n_global) return fail(ISK_ESTATE, "gather buffers not prepared");      // dp_prepare ran before the agreement
        // one group: RCCL aggregates the two small gathers into a single launch (each is latency, not bytes)
        if (g_rccl.GroupStart) RCCLCHK(g_rccl.GroupStart());
        const ncclResult_t r1 = ks64 ? g_rccl.AllGather(ks64, e->d_gks64, (size_t)e->N * d, ncclDouble, e->comm, e->stream)
                                     : g_rccl.AllGather(ks, e->d_gks, (size_t)e->N * d, ncclFloat, e->comm, e->stream);
        const ncclResult_t r2 = g_rccl.AllGather(chi, e->d_gchi, (size_t)e->N * d, ncclFloat, e->comm, e->stream);
        const ncclResult_t r3 = g_rccl.GroupEnd ? g_rccl.GroupEnd() : ncclSuccess;      // always closed, whatever r1 / r2 say
        RCCLCHK(r1); RCCLCHK(r2); RCCLCHK(r3);
        ks = e->d_gks; chi = e->d_gchi;
        if (ks64) ks64 = e->d_gks64;
    }
    return dp_isa_dev(e, ks, ks64, chi, n_global, first);
}

// a rank-local failure inside the step loop: remember the first one, raise the device flag the gradient kernels copy into the
// all-reduced slot behind the loss (TrainBuffers.rank_err) and KEEP GOING through the epoch's collectives -- the peers skip
// their updates from this step on and everybody returns at the epoch's end
static void dp_poison(isk_engine* e, int code, const std::string& msg) {
    if (e->dp_err != ISK_OK) return;
    e->dp_err = code; e->dp_errmsg = msg;
    e->h_flag[0] = 1;
    (void)hipMemcpyAsync(e->d_rank_err, e->h_flag, sizeof(int), hipMemcpyHostToDevice, e->stream);
}
// the upper bucket of a wide network's gradient: [W_2 .. b_L | batch loss | rank-error flag], contiguous behind layer 1's block
static void dp_bucket_hook(isk_engine* e) {
    const size_t first = (size_t)e->off_W[1], count = (size_t)e->nparams + 2 - first;
    (void)hipEventRecord(e->ev_b1_ready, e->stream);
    (void)hipStreamWaitEvent(e->comm_stream, e->ev_b1_ready, 0);
    e->bucket_err = (int)g_rccl.AllReduce(e->d_grad + first, e->d_grad + first, count, ncclFloat, ncclSum, e->comm, e->comm_stream);
    (void)hipEventRecord(e->ev_b1_done, e->comm_stream);
    e->bucket_issued = 1;
}
static bool dp_bucket_setup(isk_engine* e) {
    if (e->comm_stream) return true;
    if (hipStreamCreateWithFlags(&e->comm_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_b1_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_b1_done, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return true;
}

#define DPHIP(expr)                                                                                \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) dp_poison(e, ISK_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

static int dp_clear_poison(isk_engine* e) {
    const int code = e->dp_err;
    const std::string msg = e->dp_errmsg;
    e->dp_err = ISK_OK; e->dp_errmsg.clear();
    (void)hipMemsetAsync(e->d_rank_err, 0, sizeof(int), e->stream);
    (void)hipStreamSynchronize(e->stream);
    return code != ISK_OK ? fail(code, msg) : ISK_OK;
}

extern "C" int isk_dp_run(isk_engine* e, int32_t n, int32_t epochs, int32_t local_batch, int32_t nbatches, int64_t global_batch,
                          const int32_t* perms, double* losses) {
    if (!e) return fail(ISK_EINVAL, "null engine");
    const bool coll = dp_collectives(e);
    // every argument check and allocation comes BEFORE the first collective, and its outcome is agreed on by all ranks: a rank
    // that simply returned here would leave the others blocked in an all-reduce
    int lst = ISK_OK;
    std::string lmsg;
    auto bad = [&](int code, const char* msg) { if (lst == ISK_OK) { lst = code; lmsg = msg; } };
    if (need_data(e) != ISK_OK) { lst = ISK_ESTATE; lmsg = isk_last_error(); }          // a rank without data still joins the agreement
    if (n < 0 || epochs < 0) bad(ISK_EINVAL, "negative iteration count");
    if (lst == ISK_OK && (local_batch < 1 || nbatches < 1 || (long long)local_batch * nbatches > e->N || global_batch < local_batch))
        bad(ISK_EINVAL, "bad batch decomposition");
    if (perms && lst == ISK_OK)
        for (long long i = 0; i < (long long)n * epochs * e->N; ++i)
            if (perms[i] < 0 || perms[i] >= e->N) { bad(ISK_EINVAL, "permutation entry out of range"); break; }
    if (e->cfg.target_kind != ISK_TARGET_SHIFTSCALE && e->cfg.target_kind != ISK_TARGET_ISA)
        bad(ISK_EINVAL, "host-side target transform: no native data-parallel run");
    if (e->cfg.target_kind == ISK_TARGET_ISA && e->nout > ISK_ISA_DEV_MAXD && coll)
        bad(ISK_EUNSUPPORTED, "TransformISA with more than 8 outputs is solved on the host: use the host-driven data-parallel hooks");
    if (lst == ISK_OK && dp_prepare(e) != ISK_OK) { lst = ISK_ENOMEM; lmsg = isk_last_error(); }
    if (coll) CHK(dp_agree(e, lst, lmsg));
    else if (lst != ISK_OK) return fail(lst, lmsg);
    // with real collectives keep ONE gradient slab: the all-reduce is latency/link-bound, half the bytes beat the
    // few microseconds the second row slab saves in train_grad1
    const int nslab = coll ? 1 : pick_slabs(local_batch);
    // [gradient slabs | batch loss | rank-error flag]
    const size_t red_count = (size_t)(nslab - 1) * (e->nparams + 2) + (size_t)e->nparams + 2;
    // wide networks on the in-house GEMM chain: two buckets (option dp_bucket); every rank decides alike (same options, same shape)
    // single-process group with peer access: no gradient all-reduce at all (option dp_peer_reduce, the same on every engine)
    const bool peer_mode = coll && e->opt.dp_peer_reduce && e->grp && (int)e->grp->eng.size() == e->dp_world && e->dp_world > 1 && e->d_grad_base;
    e->last_run_peer = peer_mode;
    // every exit: d_grad points at the allocation's base again (the other entry points index its slabs from there), and a replica
    // that leaves a peer-reduce run early releases the group's host barrier instead of leaving the other driver threads in it
    struct Leave {
        isk_engine* e; bool peer; bool done;
        ~Leave() { if (e->d_grad_base) e->d_grad = e->d_grad_base; if (peer && !done) e->grp->leave(); }
    } leave_guard{e, peer_mode, false};
    const bool bucketed = !peer_mode && coll && e->opt.dp_bucket && e->layered && e->L >= 2 && (bf16_train_active(e) || mfma32_train_active(e)) &&
                          dp_bucket_setup(e);
    long long k = 0, step_no = 0;
    for (int it = 0; it < n; ++it) {
        if (e->cfg.featurize_every_iteration) featurize_resident(e);      // first: see isk_run
        e->side_after_forward = (long long)nbatches * epochs;
        featurize_early_share(e, e->side_after_forward);
        const int tr = dp_target_native(e);
        e->side_after_forward = 0;
        // A target that failed on THIS rank only (a HIP error in its forward, say) must not end the call here: the peers are on
        // their way into the epoch's gradient all-reduces.  Raise the flag and keep issuing them; every rank returns at the
        // epoch's end.  (With no epoch to carry the flag -- epochs == 0 -- there is nothing to keep in step: return.)
        if (tr != ISK_OK) {
            if (epochs == 0 || !coll) { (void)dp_clear_poison(e); return tr; }
            dp_poison(e, tr, isk_last_error());
        }
        // the shuffle order (drawn here from the engine's seeded generator when the host passes none: ~50 us of host
        // work) is staged while the forward runs
        if (epochs > 0) {
            const int sp = stage_perm_ahead(e, perms ? perms + k * e->N : nullptr);
            if (sp != ISK_OK) {
                if (!coll) { (void)dp_clear_poison(e); return sp; }
                dp_poison(e, sp, isk_last_error());
            }
        }
        featurize_ys_beside_training(e, (long long)nbatches * epochs);
        for (int ep = 0; ep < epochs; ++ep) {
            if (ep == 0) {
                DPHIP(hipStreamWaitEvent(e->stream, e->ev_perm, 0));
            } else {
                std::vector<int32_t> drawn;
                const int32_t* p = perms ? perms + k * e->N : nullptr;
                if (!p) { drawn.resize(e->N); shuffle_perm(e, drawn); p = drawn.data(); }
                std::memcpy(e->h_perm, p, (size_t)e->N * sizeof(int));     // validated before the first collective
                DPHIP(hipMemcpyAsync(e->d_perm, e->h_perm, (size_t)e->N * sizeof(int), hipMemcpyHostToDevice, e->stream));
            }
            if (!e->loss_acc_zeroed) DPHIP(zero_loss_acc(e));
            e->loss_acc_zeroed = false;
            {
                ScopedTimer t(e, ISK_T_TRAIN);
                const bool lazy_fold = epoch_lazy_fold(e);
                EpochGatherScope images(e, e->d_perm, local_batch, nbatches);
                for (int b = 0; b < nbatches; ++b) {
                    ++step_no;
                    if (e->opt.debug_inject_rank_error > 0 && step_no == e->opt.debug_inject_rank_error)
                        dp_poison(e, ISK_EHIP, "injected rank error (option debug_inject_rank_error)");
                    e->bucket_issued = 0; e->bucket_err = 0;
                    e->grad_hook = bucketed ? dp_bucket_hook : nullptr;
                    if (peer_mode) e->d_grad = e->d_grad_base + (size_t)(step_no & 1) * (e->nparams + 2);     // this step's slab
                    launch_grads(e, e->d_perm + (long long)b * local_batch, local_batch, 1.0f / (float)global_batch, nslab);
                    e->grad_hook = nullptr;
                    if (peer_mode) {
                        // One cross-engine dependency per step: this engine's optimizer step reads every engine's gradient slab.  The
                        // slab of step s is rewritten by step s + 2 only, which follows (on its own stream) that engine's step s + 1,
                        // which waited for every engine's gradient of s + 1, i.e. for every engine's optimizer step of s: no second wait.
                        ScopedTimer ta(e, ISK_T_ALLREDUCE);
                        const int par = (int)(step_no & 1);
                        DPHIP(hipEventRecord(e->ev_grads[par], e->stream));
                        const bool all_here = e->grp->wait();              // every engine's event is recorded before anyone waits for it
                        OptPeers pp;
                        pp.n = e->dp_world;
                        for (int r = 0; r < e->dp_world; ++r) {
                            isk_engine* o = e->grp->eng[r];
                            pp.g[r] = o->d_grad_base + (size_t)par * (o->nparams + 2);
                            if (o != e && all_here) DPHIP(hipStreamWaitEvent(e->stream, o->ev_grads[par], 0));
                        }
                        if (!all_here) dp_poison(e, ISK_EPEER, "a replica of the group left the run");
                        launch_opt(e, 1.0f, !lazy_fold || (b == nbatches - 1 && !epoch_no_fold(e)), false, b == nbatches - 1, &pp);
                        continue;
                    }
                    if (coll) {
                        ScopedTimer ta(e, ISK_T_ALLREDUCE);
                        // bucketed: the layers >= 1 (and the loss / flag slots) left from inside the backward on the second stream;
                        // what remains is layer 1's block (with the LayerNorm blocks in front of it)
                        const size_t cnt = e->bucket_issued ? (size_t)e->off_W[1] : red_count;
                        ncclResult_t r = e->bucket_issued ? (ncclResult_t)e->bucket_err : ncclSuccess;
                        if (r == ncclSuccess) r = g_rccl.AllReduce(e->d_grad, e->d_grad, cnt, ncclFloat, ncclSum, e->comm, e->stream);
                        if (e->bucket_issued) DPHIP(hipStreamWaitEvent(e->stream, e->ev_b1_done, 0));
                        // a collective this rank could not even enqueue: the peers are lost anyway, say why and stop
                        if (r != ncclSuccess) { (void)dp_clear_poison(e); return fail(ISK_EHIP, std::string("rccl: ") + g_rccl.GetErrorString(r)); }
                    }
                    e->next_idx = b + 1 < nbatches ? e->d_perm + (long long)(b + 1) * local_batch : nullptr; e->next_B = local_batch;
                    launch_opt(e, 1.0f, !lazy_fold || (b == nbatches - 1 && !epoch_no_fold(e)), false, b == nbatches - 1);
                }
            }
            double* acc = e->h_res;          // published by the last opt_step (mapped pinned memory)
            DPHIP(hipEventRecord(e->ev_loss, e->stream));
            if (ep == epochs - 1) fold_ahead(e);         // the next forward's weight images, while the host turns around (see train_epoch)
            const hipError_t se = hipEventSynchronize(e->ev_loss);
            if (se != hipSuccess) { (void)dp_clear_poison(e); return fail(ISK_EHIP, std::string("hipEventSynchronize: ") + hipGetErrorString(se)); }
            // a rank flagged an error in one of this epoch's steps: the same verdict on every rank (the flag was all-reduced)
            if (e->dp_err != ISK_OK) return dp_clear_poison(e);
            if (acc[3] != 0.0) return fail(ISK_EPEER, "a peer rank reported an error inside the training epoch; no update was applied from that step on");
            DPHIP(hipGetLastError());            // sticky error of this rank: carried by the NEXT epoch's flag slot (or returned below)
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
    leave_guard.done = true;
    if (e->dp_err != ISK_OK) return dp_clear_poison(e);       // raised behind the last epoch's collectives: nobody is left waiting
    return ISK_OK;
}

// ---------------------------------------------------------------------------------------
// one process, all GPUs of the node
// ---------------------------------------------------------------------------------------
extern "C" int isk_dp_create_all(const isk_config* cfg, const int32_t* devices, int32_t n, isk_engine** engines) {
    if (!cfg || !devices || !engines || n < 1 || n > 64) return fail(ISK_EINVAL, "bad argument");
    for (int i = 0; i < n; ++i) engines[i] = nullptr;
    CHK(rccl_load());
    if (n > 1 && !g_rccl.CommInitAll) return fail(ISK_EUNSUPPORTED, "the collective library has no ncclCommInitAll");
    int rc = ISK_OK;
    for (int i = 0; i < n && rc == ISK_OK; ++i) {
        isk_config c = *cfg;
        c.device = devices[i];
        c.seed = cfg->seed + 0x9E3779B97F4A7C15ull * (uint64_t)i;      // every rank shuffles its own shard with its own stream
        rc = isk_create(&c, &engines[i]);
    }
    if (rc == ISK_OK && n > 1) {
        std::vector<ncclComm_t> comms(n, nullptr);
        std::vector<int> devs(devices, devices + n);
        const ncclResult_t r = g_rccl.CommInitAll(comms.data(), n, devs.data());
        if (r != ncclSuccess) rc = fail(ISK_EHIP, std::string("rccl: ") + g_rccl.GetErrorString(r));
        else
            for (int i = 0; i < n; ++i) { engines[i]->comm = comms[i]; engines[i]->dp_rank = i; engines[i]->dp_world = n; }
    }
    if (rc == ISK_OK && n > 1 && n <= ISK_DP_MAXPEER) {
        // what the option dp_peer_reduce needs: every engine can read every other engine's gradient buffer (peer access between
        // distinct devices; engines on one device -- the test set-up -- share an address space anyway), and an event per step parity
        DpGroup* g = new DpGroup();
        g->eng.assign(engines, engines + n);
        bool peers_ok = true;
        for (int i = 0; i < n && peers_ok; ++i) {
            if (hipSetDevice(devices[i]) != hipSuccess) { peers_ok = false; break; }
            for (int j = 0; j < n; ++j) {
                if (devices[j] == devices[i]) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) != hipSuccess || !can) { peers_ok = false; break; }
                const hipError_t pe = hipDeviceEnablePeerAccess(devices[j], 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { peers_ok = false; break; }
                (void)hipGetLastError();
            }
            for (int k = 0; k < 2 && peers_ok; ++k)
                if (hipEventCreateWithFlags(&engines[i]->ev_grads[k], hipEventDisableTiming) != hipSuccess) peers_ok = false;
            engines[i]->d_grad_base = engines[i]->d_grad;
        }
        (void)hipGetLastError();
        if (peers_ok) for (int i = 0; i < n; ++i) engines[i]->grp = g;
        else delete g;                   // no peer access: the group keeps its RCCL all-reduce (dp_peer_reduce is then ignored)
    }
    if (rc != ISK_OK) {
        const std::string msg = isk_last_error();
        for (int i = 0; i < n; ++i) { isk_destroy(engines[i]); engines[i] = nullptr; }
        return fail(rc, msg);
    }
    return ISK_OK;
}

extern "C" void isk_dp_destroy_all(isk_engine** engines, int32_t n) {
    if (!engines) return;
    for (int i = 0; i < n; ++i) { isk_destroy(engines[i]); engines[i] = nullptr; }
}

// one host thread per engine for the duration of the call; returns the first failing rank's status and message
template <class F>
static int for_each_engine(isk_engine** engines, int32_t n, F&& body) {
    std::vector<int> rc(n, ISK_OK);
    std::vector<std::string> msg(n);
    auto work = [&](int i) {
        rc[i] = body(i);
        if (rc[i] != ISK_OK) msg[i] = isk_last_error();            // thread-local: copied out before the thread ends
    };
    if (n == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < n; ++i) th.emplace_back(work, i);
        for (auto& t : th) t.join();
    }
    // a rank's OWN error explains more than the ISK_EPEER it causes on the others
    for (int i = 0; i < n; ++i)
        if (rc[i] != ISK_OK && rc[i] != ISK_EPEER) return fail(rc[i], "rank " + std::to_string(i) + ": " + msg[i]);
    for (int i = 0; i < n; ++i)
        if (rc[i] != ISK_OK) return fail(rc[i], "rank " + std::to_string(i) + ": " + msg[i]);
    return ISK_OK;
}

extern "C" int isk_dp_set_data_all(isk_engine** engines, int32_t n, const float* xs, const float* ys, int64_t D, int64_t K, int64_t N) {
    if (!engines || n < 1 || !xs || !ys) return fail(ISK_EINVAL, "null argument");
    if (N < n || N % n != 0) return fail(ISK_EINVAL, "the native data-parallel run needs equal shards: N must be a multiple of the engine count");
    const int64_t Nl = N / n;
    // N is the slowest axis of xs (D, N) and ys (D, K, N): rank r's slab is contiguous (src/data.jl:9-11)
    return for_each_engine(engines, n, [&](int i) {
        return isk_set_data(engines[i], xs + (size_t)i * Nl * D, ys + (size_t)i * Nl * K * D, D, K, Nl);
    });
}

extern "C" int isk_dp_set_params_all(isk_engine** engines, int32_t n, const float* flat, int64_t nflat) {
    if (!engines || n < 1 || !flat) return fail(ISK_EINVAL, "null argument");
    for (int i = 0; i < n; ++i) CHK(isk_set_params(engines[i], flat, nflat));
    return ISK_OK;
}

extern "C" int isk_dp_run_all(isk_engine** engines, int32_t n_engines, int32_t n, int32_t epochs, int32_t minibatch,
                              const int32_t* const* perms, double* losses) {
    if (!engines || n_engines < 1) return fail(ISK_EINVAL, "null argument");
    for (int i = 0; i < n_engines; ++i)
        if (!engines[i] || engines[i]->N == 0) return fail(ISK_ESTATE, "no data: call isk_dp_set_data_all first");
    // train_batch!'s batch rule on the GLOBAL data (src/iso.jl:180-181): a global batch is the union of one slice per rank
    const long long Nl = engines[0]->N, Ng = Nl * n_engines;
    const long long mb = minibatch < 0 ? engines[0]->cfg.minibatch : minibatch;
    const long long bs = (mb == 0 || Ng < mb) ? Ng : mb;
    long long bl, nb, gb;
    if (bs == Ng) { bl = Nl; nb = 1; gb = Ng; }
    else {
        // train_batch!'s rule (src/iso.jl:180) fixes batch size and step count; a global batch that does not split evenly would
        // silently become bl * n_engines rows and change both
        if (bs % n_engines != 0) return fail(ISK_EINVAL, "the global minibatch must be a multiple of the engine count");
        bl = bs / n_engines;
        if (bl < 1) return fail(ISK_EINVAL, "global minibatch smaller than the number of engines");
        nb = Nl / bl; gb = bl * n_engines;
    }
    if (engines[0]->grp) engines[0]->grp->reset();
    std::vector<std::vector<double>> ls(n_engines, std::vector<double>((size_t)std::max(1, n * epochs), 0.0));
    const int rc = for_each_engine(engines, n_engines, [&](int i) {
        return isk_dp_run(engines[i], n, epochs, (int32_t)bl, (int32_t)nb, gb, perms ? perms[i] : nullptr, ls[i].data());
    });
    if (rc != ISK_OK) return rc;
    if (losses)
        for (int k = 0; k < n * epochs; ++k) losses[k] = ls[0][k] / (double)Ng;      // train_batch!'s return value (src/iso.jl:193)
    return ISK_OK;
}
