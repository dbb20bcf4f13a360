/*
 * isokann_hip.h -- C ABI of the MI355X-native ISOKANN training engine.
 *
 * Drop-in boundary for the per-iteration hot path of axsk/ISOKANN.jl (v2.0.0):
 * featurize -> MLP forward over the D x K x N Koopman samples -> K-mean ->
 * shift-scale / ISA target -> MLP forward/backward + optimizer on xs.
 *
 * The reference has NO FFI/plugin interface on this path (it is Julia multiple
 * dispatch on a Flux.Chain); each entry point below names the reference function
 * (file:line, relative to the reference repo) whose role it takes when a Julia
 * host binds this library with `ccall` (see INTEGRATION.md for the Julia shim).
 *
 * Conventions
 *  - Every array is Julia column-major, passed as a plain pointer + sizes:
 *      xs (D, N), ys (D, K, N), model outputs (nout, N), weights W (out, in).
 *    In C terms xs is [N][D], ys is [N][K][D]: one sample's D values contiguous.
 *  - The flat parameter vector is Flux's trainables order
 *      [LayerNorm.scale, LayerNorm.bias, W1, b1, W2, b2, ...], W column-major.
 *  - Caller owns every host buffer.  The engine copies on set_* and writes results
 *    into caller-allocated buffers; it owns all device memory.
 *  - Every function returns an isk_status; nothing throws across the ABI.
 *    isk_last_error() gives a thread-local message for the last failure.
 *  - An engine handle is not thread-safe; distinct handles are independent.
 *  - Pointers are HOST pointers unless the parameter name ends in `_dev`.
 */
#ifndef ISOKANN_HIP_H
#define ISOKANN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISK_MAX_LAYERS 8
#define ISK_ABI_VERSION 2

typedef struct isk_engine isk_engine;

typedef enum isk_status {
    ISK_OK = 0,
    ISK_EINVAL = 1,        /* bad shape / argument (the reference's @assert failures, isotarget.jl:37,102) */
    ISK_EHIP = 2,          /* HIP runtime error */
    ISK_ENOMEM = 3,
    ISK_ECONSTANT_CHI = 4, /* DomainError "chi function is constant"          src/isotarget.jl:39     */
    ISK_ECOLLAPSED = 5,    /* DomainError "model collapsed under training"    src/iso.jl:186-189      */
    ISK_ESINGULAR = 6,     /* DomainError "simplex transformation ... singular" src/isotarget.jl:94-97 */
    ISK_EUNSUPPORTED = 7,  /* configuration outside what the kernels cover */
    ISK_ESTATE = 8,        /* call order (e.g. train before set_data) */
    ISK_EPEER = 9          /* data-parallel run: ANOTHER rank failed (its own call returns the cause); this rank stopped with it */
} isk_status;

typedef enum { ISK_ACT_IDENTITY = 0, ISK_ACT_SIGMOID = 1, ISK_ACT_TANH = 2, ISK_ACT_RELU = 3, ISK_ACT_SWISH = 4 } isk_act;
typedef enum { ISK_FEAT_IDENTITY = 0, ISK_FEAT_ALLPAIRS = 1, ISK_FEAT_PAIRLIST = 2 } isk_featurizer;
typedef enum { ISK_OPT_ADAM = 0, ISK_OPT_NESTEROV = 1 } isk_opt;
/* ISK_TARGET_HOST: the host computes the target from isk_chis / isk_koopman and passes it to isk_train_batch
 * (the reference's experimental transforms, e.g. TransformPseudoInv src/isotarget.jl:145-179, whose small d x d
 * LAPACK steps run on the CPU in the reference as well); isk_target / isk_run then return ISK_EINVAL */
typedef enum { ISK_TARGET_SHIFTSCALE = 0, ISK_TARGET_ISA = 1, ISK_TARGET_HOST = 2 } isk_target_kind;
/* Flux `normalise` epsilon placement: 0 = (x-mu)/sqrt(var+eps^2) [Flux >= 0.14, default],
 * 1 = (x-mu)/(std+eps) [Flux <= 0.13], 2 = (x-mu)/sqrt(var+eps) [PyTorch] */
typedef enum { ISK_LN_VAR_EPS2 = 0, ISK_LN_STD_EPS = 1, ISK_LN_VAR_EPS = 2 } isk_ln_mode;

/* Mirrors the keyword arguments of Iso(...) (src/iso.jl:17-25), pairnet/densenet
 * (src/models.jl:65,87), the optimizer constructors (src/models.jl:12,20) and the
 * featurizer structs (src/utils/features.jl:22-35). */
typedef struct isk_config {
    int32_t abi_version;                 /* ISK_ABI_VERSION */
    int32_t device;                      /* HIP device ordinal */
    /* model: densenet(layers = widths[0..n_layers]) */
    int32_t n_layers;                    /* number of Dense layers, 1..ISK_MAX_LAYERS */
    int32_t widths[ISK_MAX_LAYERS + 1];  /* in, hidden..., nout */
    int32_t activation;                  /* isk_act, hidden layers (Flux.sigmoid default) */
    int32_t lastactivation;              /* isk_act, last layer (identity default) */
    int32_t layernorm;                   /* leading Flux.LayerNorm(widths[0]) */
    int32_t ln_mode;                     /* isk_ln_mode */
    float   ln_eps;                      /* 1e-5 */
    /* optimizer: OptimiserChain(WeightDecay(reg), Adam(lr) | Nesterov(lr, rho)) */
    int32_t opt_kind;                    /* isk_opt */
    float   lr, reg, beta1, beta2, opt_eps, rho;
    /* featurizer */
    int32_t featurizer;                  /* isk_featurizer */
    int32_t n_atoms;                     /* coords have 3*n_atoms rows (ALLPAIRS / PAIRLIST) */
    int32_t n_pairs;                     /* PAIRLIST: number of pairs (= widths[0]) */
    const int32_t* pairs;                /* PAIRLIST: 2*n_pairs 0-based atom indices (a0,b0,a1,b1,...);
                                            copied by isk_create.  ALLPAIRS over an atom subset
                                            (FeaturesAtoms) is expressed as a PAIRLIST by the host. */
    int32_t featurize_every_iteration;   /* 1: pdist runs inside every isk_run iteration (north-star
                                            hot path); 0: features cached at isk_set_data, as the
                                            reference does (src/simulation.jl:110-114) */
    /* target transform */
    int32_t target_kind;                 /* isk_target_kind; ISA needs nout > 1 */
    int32_t isa_permute;                 /* TransformISA.permute  (default 1) */
    int32_t isa_whitening;               /* TransformISA.whitening (default 0) */
    int32_t minibatch;                   /* Iso.minibatch (default 100; 0 = full batch) */
    uint64_t seed;                       /* shuffle seed for DataLoader(shuffle=true) when no
                                            permutation is injected */
    int32_t forward_mode;                /* arithmetic of the layer-1 GEMM in the forward over the Koopman
                                            samples: 0 = f32-input MFMA (exact f32 fma chain),
                                            1 = split-bf16: three bf16 planes per operand, six bf16 MFMAs per
                                            product block, f32 accumulate (f32-equivalent accuracy at 2.7x the
                                            f32 MFMA rate),
                                            2 = plain bf16 operands (leading plane only), f32 accumulate: the C5
                                            configuration of the baseline; outputs agree with f32 to ~1e-2 relative,
                                            3 = split-f16: two scaled f16 planes per operand (22 mantissa bits; the
                                            samples are written pre-split by the featurizer, the weights at the fold),
                                            three f16 MFMAs per product block, f32 accumulate: f32-equivalent accuracy,
                                            half the matrix work of mode 1 and no conversion in the GEMM loop.  Shapes
                                            it does not cover (layered path, > 2048 features, > 682 atoms) run mode 1
                                            (isk_get_option "forward_mode" reports the effective mode).
                                            Training kernels are f32 in every mode (wide networks: option train_bf16). */
    int32_t force_layered;               /* 1: run this network on the layered (wide-network) path even if the fused kernels
                                            could hold it (test coverage of that path); 0 = by shape */
} isk_config;

/* ---- library ------------------------------------------------------------------ */
int  isk_abi_version(void);
int  isk_device_count(int32_t* count);
const char* isk_last_error(void);
const char* isk_status_string(int status);
void isk_config_default(isk_config* cfg);           /* the reference's defaults */

/* ---- lifetime: Iso(...) / gpu(iso)   src/iso.jl:17-43, 256 -------------------- */
int  isk_create(const isk_config* cfg, isk_engine** out);
void isk_destroy(isk_engine* e);
int  isk_synchronize(isk_engine* e);
int  isk_set_stream(isk_engine* e, void* hip_stream);  /* run on the caller's stream (e.g. torch's) */

/* ---- options: per-engine knobs that select a kernel variant or the schedule of an iteration --------------------------
 * (no counterpart in the reference: Flux picks its kernels by dispatch).  Integer-valued; isk_create takes the defaults
 * below, or ISK_<KEY> from the environment of the creating process; nothing is latched process-wide.
 *   key                 default  meaning
 *   overlap_featurize   2        where the NEXT iteration's featurization runs: 2 = side stream beside the training kernels,
 *                                1 = beside the forward, 0 = inline on the main stream
 *   side_min_steps      4        fewest optimizer steps of the coming training phase for which the side stream is used
 *   feat_wg_per_cu      1        throttle of the background featurizer (workgroups per CU)
 *   feat_bg_wgs         0        ... or its workgroups in total (0 = the per-CU rule)
 *   feat_pf             1        samples in flight per wave of the background featurizer (1 | 2)
 *   train_graph         1        replay a training epoch as one hipGraph
 *   grad1_nw8           -1       layer-1 gradient kernel: 1 = 8-wave variant, 0 = 16-wave, -1 = 8-wave while the featurizer runs
 *   tail_narrow         1        training: register-resident tail kernel where every tail layer is <= 16 wide (0: the generic LDS kernel)
 *   fuse_opt            1        training on one GPU with batches of at most 768 rows (one row slab): the optimizer step rides on the
 *                                layer-1 gradient kernel -- three launches per optimizer step instead of four
 *   dense_groups        1        wide networks, bf16 forward: one grouped launch per layer
 *   wide_fused          1        wide networks, plain-bf16 forward (forward_mode 2): one fused kernel, activations in LDS
 *   train_bf16          0        wide networks: training GEMMs on bf16 operands (f32 master weights, f32 accumulate); 0 = the same
 *                                in-house MFMA kernels in Float32 arithmetic (no vendor BLAS is reachable from this library)
 *   fuse_head           1        wide networks with at least three layers and at most four outputs, training GEMMs: the last layer's forward,
 *                                the loss, delta_L and delta_{L-1} are ONE launch (three before); its per-workgroup partial sums of
 *                                dW_L / db_L / the loss are added up by the next launch
 *   fuse_boundary       1        wide networks without a leading LayerNorm, training GEMMs: the optimizer step, the next step's weight images
 *                                and the next step's batch images are ONE launch at the boundary between two optimizer steps
 *   dp_bucket           1        native data-parallel run of a wide network: the gradient all-reduce in two buckets -- the blocks of the
 *                                layers >= 1 travel on a second stream while the first layer's backward products run (bit-identical results)
 *   dp_peer_reduce      0        isk_dp_run_all (one process, several GPUs): 1 = the gradient is not all-reduced; every engine's optimizer
 *                                step reads the other engines' gradient buffers over peer access and sums them in rank order
 *   dp_force_coll       0        issue the native collectives on a one-rank communicator (tests, bench.py --shard-of)
 *   debug_inject_rank_error 0    test hook: this rank flags a local failure at its n-th optimizer step of a data-parallel run
 * (18 keys.  A library built with -DISK_EXPERIMENTAL additionally carries the knobs of variants that were measured and rejected --
 * train_persist, fuse_tail, epoch_gather, fuse_opt_rows, g16_waves, feat_early_pct / feat_early_wgs, side_early, bf16_skew,
 * fwd_waves_per_simd -- and the diagnostics debug_feat_copy / debug_stamps / debug_rowmod, with the kernels behind them; the shipped
 * library has neither: DESIGN.md section 6b.)
 * Read-only keys of isk_get_option: forward_mode (effective), layered, planes, nparams_engine, wide_fwd (1 when the resident
 * data set runs the fused bf16 forward of wide networks: layered network, forward_mode 2, pair features, option wide_fused),
 * dp_peer_active (1 when the last isk_dp_run / isk_dp_run_all summed the gradients over peer access). */
int  isk_set_option(isk_engine* e, const char* key, int64_t value);
int  isk_get_option(isk_engine* e, const char* key, int64_t* value);

/* ---- parameters / optimizer state: Flux.setup, cpu(iso), save/load
 *      src/iso.jl:27,74,257,405-420 ----------------------------------------------- */
int  isk_nparams(isk_engine* e, int64_t* n);
int  isk_set_params(isk_engine* e, const float* flat, int64_t n);
int  isk_get_params(isk_engine* e, float* flat, int64_t n);
/* Adam: m = first moment, v = second moment, bt = (beta1^t, beta2^t).
 * Nesterov: m = velocity, v ignored (may be NULL). */
int  isk_set_opt_state(isk_engine* e, const float* m, const float* v, int64_t n, const double bt[2]);
int  isk_get_opt_state(isk_engine* e, float* m, float* v, int64_t n, double bt[2]);
int  isk_reset_opt_state(isk_engine* e);            /* Flux.setup(opt, model) */

/* ---- data: SimulationData(xs, ys; featurizer)   src/simulation.jl:100-114 ------ */
/* xs (D, N), ys (D, K, N).  D = 3*n_atoms for pdist featurizers, widths[0] for
 * identity.  Uploads, featurizes (pdist) and LayerNorm-standardises once. */
int  isk_set_data(isk_engine* e, const float* xs, const float* ys, int64_t D, int64_t K, int64_t N);
int  isk_set_data_dev(isk_engine* e, const float* xs_dev, const float* ys_dev, int64_t D, int64_t K, int64_t N);
/* Data growth of adaptive sampling without re-uploading what is already resident: `iso.data = mergedata(iso.data, new)` of
 * addcoords! (src/iso.jl:238, src/simulation.jl:162-185) and the `cutoff` window `iso.data = iso.data[end-cutoff+1:end]` of
 * run_kde! (src/iso.jl:288-290).  Drops the first `drop_first` resident start points (with their K samples), then appends
 * n_add new ones (xs (D, n_add), ys (D, K, n_add); NULL / 0: none).  Only the new columns cross PCIe; the resident ones move
 * device-to-device.  Model, optimizer state and options stay; Koopman weights (isk_set_koopman_weights) are dropped as by
 * isk_set_data.  ISK_ESTATE without resident data, ISK_EINVAL when nothing would be left or K / D differ. */
int  isk_append_data(isk_engine* e, const float* xs, const float* ys, int64_t D, int64_t K, int64_t n_add, int64_t drop_first);
int  isk_data_shape(isk_engine* e, int64_t* D, int64_t* K, int64_t* N);
/* WeightedSamples (Girsanov reweighting), src/data.jl:187-215: expectation becomes sum_k w[k,n] f(y[k,n]) / K.
 * w is (1, K, N) column-major, i.e. [N][K]; NULL restores the plain K-mean. */
int  isk_set_koopman_weights(isk_engine* e, const float* w, int64_t K, int64_t N);

/* ---- featurizer: f(coords)   src/utils/features.jl:22-35, pairdists.jl:6-24,109-127 */
/* coords (D, ncols) -> out (F, ncols) */
int  isk_features(isk_engine* e, const float* coords, int64_t D, int64_t ncols, float* out);

/* ---- model call: iso.model(x)   src/iso.jl:185,203; src/isotarget.jl:18,101 ---- */
/* x (F, ncols) features -> out (nout, ncols) */
int  isk_forward(isk_engine* e, const float* x, int64_t F, int64_t ncols, float* out);
/* chicoords(iso, coords)   src/iso.jl:211 : coords (D, ncols) -> (nout, ncols) */
int  isk_chicoords(isk_engine* e, const float* coords, int64_t D, int64_t ncols, float* out);
/* chis(iso)   src/iso.jl:203 : (nout, N) on the resident xs */
int  isk_chis(isk_engine* e, float* out);
/* koopman(iso) = expectation(model, propfeatures)   src/isotarget.jl:18,20 : (nout, N) */
int  isk_koopman(isk_engine* e, float* out);
/* The same K-means before their rounding to Float32: the sums of the K Float32 model outputs taken in Float64.  This is what the
 * engine's TransformISA solve reads (the reference rounds K chi to the model's eltype first, src/isotarget.jl:18,103, and widens
 * it again in myisa, :82; that one rounding, amplified by the simplex inverse, is most of an fp32 run's distance from a Float64
 * trajectory -- DESIGN.md section 5).  Float64(isk_koopman) wherever the engine keeps no double copy. */
int  isk_koopman_f64(isk_engine* e, double* out);

/* ---- evaluation callers on the same kernels ----------- */
/* expectation(model, featurizer(ys)) for caller-provided coordinates ys (D, K, ncols) -> (nout, ncols)
 * src/isotarget.jl:18 on data that is not the resident set (validation data, resampling) */
int  isk_koopman_coords(isk_engine* e, const float* ys, int64_t D, int64_t K, int64_t ncols, float* out);
/* validationloss(iso, valdata)   src/iso.jl:160-168 : mean |chi(vx) - shiftscale([K chi(vy); K chi(ys)])|^2,
 * 1-D chi only; vxs (D, Nv), vys (D, K, Nv) are validation COORDINATES */
int  isk_validationloss(isk_engine* e, const float* vxs, const float* vys, int64_t D, int64_t K, int64_t Nv, double* loss);

/* ---- d chi / d coords ----------------------------------- */
/* Zygote-style pullback of chicoords: grad (D, ncols) = (d chi(coords_n) / d coords_n)' * cot[:, n], through
 * the MLP, the LayerNorm and the pairwise-distance featurizer.  Counterpart of the reference's
 * sqpairdist_bwd_kernel! + rrule (src/utils/pairdists.jl:153-196); consumers: src/utils/minimumpath.jl:3-13,
 * src/simulators/metadynamics.jl:40-43, src/simulators/openmm.jl:659-664.  cot is (nout, ncols).
 * Covers every shape isk_create accepts (round 5: also beyond 2048 features / 682 atoms). */
int  isk_chicoords_pullback(isk_engine* e, const float* coords, int64_t D, int64_t ncols, const float* cot, float* grad);

/* ---- target: isotarget(iso)   src/isotarget.jl:10-42, 74-127 ------------------- */
/* Computes the target on the resident data, keeps it on the device for training and,
 * if out != NULL, copies it to out (nout, N). */
int  isk_target(isk_engine* e, float* out);

/* ---- training: train_batch!   src/iso.jl:179-194 ------------------------------- */
/* One epoch.  target (nout, N) host array, or NULL to use the device-resident one from
 * the last isk_target.  perm: N 0-based indices (the DataLoader shuffle order) or NULL
 * for the engine's own seeded shuffle.  minibatch < 0 uses cfg.minibatch.
 * Returns sum-of-batch-losses / N in *loss (Float64, src/iso.jl:193). */
int  isk_train_batch(isk_engine* e, const float* target, int32_t minibatch, const int32_t* perm, double* loss);
/* Gradient of  l / numobs  for one batch of column indices, Flux trainables order
 * (what Zygote hands to update!, src/iso.jl:184-191).  No parameter update.
 * l (the un-normalised batch loss) is returned in *loss. */
int  isk_batch_grads(isk_engine* e, const float* target, const int32_t* idx, int32_t nidx, float* grads, double* loss);

/* ---- run!(iso, n, epochs)   src/iso.jl:72-94 ----------------------------------- */
/* perms: n*epochs permutations of N (consumed in order) or NULL.  losses: n*epochs. */
int  isk_run(isk_engine* e, int32_t n, int32_t epochs, const int32_t* perms, double* losses);
/* iso.minibatch = mb on the reference's mutable Iso (src/iso.jl:10): the batch size isk_run uses from now on (0 = full batch) */
int  isk_set_minibatch(isk_engine* e, int32_t minibatch);

/* ---- data-parallel hooks (one process per GPU; collectives stay with the host) -- */
/* local (unreduced) statistics the target needs across shards */
int  isk_dp_koopman_local(isk_engine* e);                       /* ks for the local shard, stays on device */
int  isk_dp_minmax_local(isk_engine* e, float mm[2]);           /* local min, max of ks (1-D) */
int  isk_dp_shiftscale_apply(isk_engine* e, float gmin, float gmax);   /* target = (ks-gmin)/(gmax-gmin) */
int  isk_dp_get_ks_chi(isk_engine* e, float* ks, float* chi);   /* local (nout, N) slices for the ISA gather */
int  isk_dp_isa_apply(isk_engine* e, const double* At, const int32_t* rowperm); /* target = (A' ks)[perm] */
int  isk_dp_set_loss_weights(isk_engine* e, const float* w);    /* global 1/std(target) */
/* Device-resident ISA for data-parallel runs: the host all-gathers the local slices (device pointers
 * from isk_dp_ks_chi_dev, `nfloats` each) into ks_all/chi_all ([n_global][d], rank order) and every
 * rank runs the same Float64 solve on them, writing the target rows [first, first+N_local) of its own
 * shard and the global loss weights.  No host synchronisation; isk_dp_check_target reports the
 * DomainError status (one sync), normally folded into isk_dp_epoch_end by the caller. */
int  isk_dp_ks_chi_dev(isk_engine* e, void** ks_dev, void** chi_dev, int64_t* nfloats);
int  isk_dp_isa_dev(isk_engine* e, const float* ks_all_dev, const float* chi_all_dev, int64_t n_global, int64_t first);
int  isk_dp_check_target(isk_engine* e);
/* per-row sum and sum of squares of the local target (for the global std, src/iso.jl:183) */
int  isk_dp_target_moments(isk_engine* e, double* sum, double* sumsq);
/* Host Float64 ISA solve on gathered arrays (no device work): ks, chi are (d, N) column-major.
 * Outputs At (d*d, row-major: target = At * ks) and the fixperm row permutation.
 * myisa + fixperm, src/isotarget.jl:81-98, 120-127. */
int  isk_isa_solve(const float* ks, const float* chi, int64_t N, int32_t d, int32_t permute, int32_t whitening,
                   double* At, int32_t* rowperm);
/* device pointer + length of the flat gradient/loss buffer: [grads (nparams) | batch loss] */
int  isk_dp_grad_buffer(isk_engine* e, void** buf_dev, int64_t* nfloats);
/* Epoch protocol (no host synchronisation inside the step loop):
 *   isk_dp_epoch_begin(perm, n)   upload this rank's shuffle of its shard, zero the loss accumulator
 *   per optimizer step:
 *     isk_dp_local_grads(first, nidx, global_batch)   rows perm[first .. first+nidx): this rank's
 *         contribution to the gradient of (global batch loss / global_batch) + its batch loss -> buffer
 *     <host all-reduces (sum) the buffer on the engine's stream>
 *     isk_dp_apply()              optimizer step from the reduced buffer; a non-finite reduced loss
 *                                 skips this and every later update of the epoch
 *   isk_dp_epoch_end(&loss_sum)   one synchronisation: sum of the reduced batch losses, or
 *                                 ISK_ECOLLAPSED */
int  isk_dp_epoch_begin(isk_engine* e, const int32_t* perm, int32_t n);
int  isk_dp_local_grads(isk_engine* e, int32_t first, int32_t nidx, int64_t global_batch);
int  isk_dp_apply(isk_engine* e);
int  isk_dp_epoch_end(isk_engine* e, double* loss_sum);

/* Native collectives: the same data-parallel run! with RCCL called from the engine on its own stream, so an
 * iteration has no host code between the kernels of a step (run! src/iso.jl:72-94 across ranks; the hooks above
 * remain for hosts that bring their own collectives, e.g. MPI.jl or a CPU gloo group).
 *   isk_dp_comm_id(id)                     rank 0: ncclGetUniqueId; the host broadcasts the 128 bytes
 *   isk_dp_comm_init(e, id, rank, world)   every rank: ncclCommInitRank (librccl.so is loaded on first use)
 *   isk_dp_run(e, n, epochs, local_batch, nbatches, global_batch, perms, losses)
 *       per iteration: K-mean forward on the shard; 1-D target: ONE all-reduce(max) of [-min, max];
 *       ISA: all-gather of the ks / chi slices, every rank runs the Float64 simplex kernels on the gathered arrays;
 *       per optimizer step: ONE in-place all-reduce(sum) of the flat [gradient slabs | batch loss] buffer.
 *       Shards must be equal (N the same on every rank).  perms: (n*epochs, N) shuffles of the LOCAL shard, of
 *       which the first nbatches*local_batch entries are used (NULL: the engine draws them from its seeded generator,
 *       like isk_run, while the forward runs); losses[k] = sum of the reduced batch losses of
 *       epoch k (divide by the global N for train_batch!'s return value, src/iso.jl:193).
 * Without isk_dp_comm_init (or with world == 1) isk_dp_run runs the same code without collectives.
 * Errors are agreed on: argument / allocation failures of any rank by one small all-reduce before the run's first collective,
 * failures inside the step loop through a flag slot behind the batch loss in every gradient all-reduce -- every rank skips the
 * update in that same step and returns (the failing rank its own status, the others ISK_EPEER) instead of waiting for a peer. */
int  isk_dp_comm_id(uint8_t id[128]);
int  isk_dp_comm_init(isk_engine* e, const uint8_t id[128], int32_t rank, int32_t world);
int  isk_dp_run(isk_engine* e, int32_t n, int32_t epochs, int32_t local_batch, int32_t nbatches, int64_t global_batch,
                const int32_t* perms, double* losses);

/* ONE host process for all GPUs of a node -- the reference is a single Julia process that calls gpu(iso) once (src/iso.jl:256):
 *   isk_dp_create_all(cfg, devices, n, engines)    n engines (cfg->device is replaced by devices[i]) + ONE communicator group
 *                                                  (ncclCommInitAll); engines[] are ordinary handles (isk_chis, isk_get_params, ...)
 *   isk_dp_set_params_all(engines, n, flat, len)   the same Flux-order parameter vector on every replica
 *   isk_dp_set_data_all(engines, n, xs, ys, D, K, N)   SimulationData(xs, ys): engine i receives the contiguous N-slab
 *                                                  [i N/n, (i+1) N/n) of xs (D, N) and ys (D, K, N); N must be a multiple of n
 *   isk_dp_run_all(engines, n_eng, n, epochs, minibatch, perms, losses)   run!(iso, n, epochs) on the sharded data: one host
 *       thread per engine inside the call drives isk_dp_run (the step loop is launch-bound; the Julia host sees one blocking
 *       ccall).  minibatch keeps its meaning on the GLOBAL data (src/iso.jl:180; < 0: cfg.minibatch) and must be a multiple of
 *       n_eng (ISK_EINVAL otherwise: rounding it would change train_batch!'s batch size and step count); perms: NULL, or one
 *       (n*epochs, N/n) array of LOCAL shuffles per engine; losses[k] = train_batch!'s return value (src/iso.jl:193)
 *   isk_dp_destroy_all(engines, n) */
int  isk_dp_create_all(const isk_config* cfg, const int32_t* devices, int32_t n, isk_engine** engines);
int  isk_dp_set_params_all(isk_engine** engines, int32_t n, const float* flat, int64_t nflat);
int  isk_dp_set_data_all(isk_engine** engines, int32_t n, const float* xs, const float* ys, int64_t D, int64_t K, int64_t N);
int  isk_dp_run_all(isk_engine** engines, int32_t n_engines, int32_t n, int32_t epochs, int32_t minibatch,
                    const int32_t* const* perms, double* losses);
void isk_dp_destroy_all(isk_engine** engines, int32_t n);

/* ---- instrumentation -------------------------------------------------------------- */
/* ISK_T_ALLREDUCE: the gradient all-reduces of the native data-parallel run (inside the ISK_T_TRAIN span);
 * ISK_T_TARGET includes the target's own collectives (min/max all-reduce, ks / chi all-gather) */
typedef enum { ISK_T_FEATURIZE = 0, ISK_T_FWD_YS = 1, ISK_T_FWD_XS = 2, ISK_T_TARGET = 3,
               ISK_T_TRAIN = 4, ISK_T_ALLREDUCE = 5, ISK_T_COUNT = 6 } isk_timer_id;
/* HIP-event timing of the named phase on the engine's stream (enable before the run) */
int  isk_timers_enable(isk_engine* e, int32_t on);
int  isk_timers_read(isk_engine* e, double total_ms[ISK_T_COUNT], int64_t count[ISK_T_COUNT]);
int  isk_timers_reset(isk_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* ISOKANN_HIP_H */
