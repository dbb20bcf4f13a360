// Per-engine options: every knob that selects a kernel variant or the schedule of an iteration (IskOptions, isk_internal.h).
// A Julia host sets them per engine through the C ABI; the environment variable ISK_<KEY> of the process that calls
// isk_create only provides the default of a NEW engine (bench / test sweeps) -- no value is latched in a static.
#include "engine.hpp"

namespace {
struct OptDesc {
    const char* key;
    int IskOptions::*field;
    int lo, hi;
};
const OptDesc kOptions[] = {
    {"overlap_featurize", &IskOptions::overlap_featurize, 0, 2},
    {"side_min_steps", &IskOptions::side_min_steps, 0, 1 << 30},
    {"feat_wg_per_cu", &IskOptions::feat_wg_per_cu, 1, 16},
    {"feat_bg_wgs", &IskOptions::feat_bg_wgs, 0, 65536},
    {"feat_pf", &IskOptions::feat_pf, 1, 2},
    {"dense_groups", &IskOptions::dense_groups, 0, 1},
    {"train_graph", &IskOptions::train_graph, 0, 1},
    {"grad1_nw8", &IskOptions::grad1_nw8, -1, 1},
    {"dp_force_coll", &IskOptions::dp_force_coll, 0, 1},
    {"dp_bucket", &IskOptions::dp_bucket, 0, 1},
    {"dp_peer_reduce", &IskOptions::dp_peer_reduce, 0, 1},
    {"train_bf16", &IskOptions::train_bf16, 0, 1},
    {"tail_narrow", &IskOptions::tail_narrow, 0, 1},
    {"fuse_opt", &IskOptions::fuse_opt, 0, 1},
    {"fuse_boundary", &IskOptions::fuse_boundary, 0, 1},
    {"fuse_head", &IskOptions::fuse_head, 0, 1},
    {"wide_fused", &IskOptions::wide_fused, 0, 1},
    {"debug_inject_rank_error", &IskOptions::debug_inject_rank_error, 0, 1 << 30},
#ifdef ISK_EXPERIMENTAL
    // Knobs of measured-and-rejected variants and of diagnostics.  A library built with -DISK_EXPERIMENTAL
    // (scripts/devbuild.sh, DEVFLAGS_ALL) carries them and the kernels behind them; the shipped library has the defaults only.
    {"bf16_skew", &IskOptions::bf16_skew, 0, 1},
    {"side_early", &IskOptions::side_early, 0, 1},
    {"feat_early_pct", &IskOptions::feat_early_pct, 0, 100},
    {"feat_early_wgs", &IskOptions::feat_early_wgs, 2, 4096},
    {"fwd_waves_per_simd", &IskOptions::fwd_waves_per_simd, 1, 4},
    {"fuse_tail", &IskOptions::fuse_tail, 0, 1},
    {"g16_waves", &IskOptions::g16_waves, 0, 8},
    {"epoch_gather", &IskOptions::epoch_gather, 0, 1},
    {"fuse_opt_rows", &IskOptions::fuse_opt_rows, 0, 1024},
    {"train_persist", &IskOptions::train_persist, 0, 2},
    {"debug_feat_copy", &IskOptions::debug_feat_copy, 0, 64},
    {"debug_stamps", &IskOptions::debug_stamps, 0, 1},
    {"debug_rowmod", &IskOptions::debug_rowmod, 0, 1 << 30},
#endif
};

const OptDesc* find_option(const char* key) {
    for (const OptDesc& d : kOptions)
        if (std::strcmp(d.key, key) == 0) return &d;
    return nullptr;
}
}  // namespace

#ifdef ISK_EXPERIMENTAL
extern "C" int isk_experimental_build(void) { return 1; }       // (not part of the C ABI: lets the test-suite see which library it loaded)
#endif

void engine_options_init(isk_engine* e) {
    e->opt = IskOptions();
    for (const OptDesc& d : kOptions) {
        std::string env = "ISK_";
        for (const char* c = d.key; *c; ++c) env += (char)std::toupper((unsigned char)*c);
        const char* v = getenv(env.c_str());
        if (!v || !*v) continue;
        const long x = std::strtol(v, nullptr, 10);
        if (x >= d.lo && x <= d.hi) e->opt.*(d.field) = (int)x;
    }
}

extern "C" int isk_set_option(isk_engine* e, const char* key, int64_t value) {
    if (!e || !key) return fail(ISK_EINVAL, "null argument");
    const OptDesc* d = find_option(key);
    if (!d) return fail(ISK_EINVAL, std::string("unknown option: ") + key);
    if (value < d->lo || value > d->hi) return fail(ISK_EINVAL, std::string("option value out of range: ") + key);
    if (e->opt.*(d->field) == (int)value) return ISK_OK;
    // an option may change what a captured epoch graph or an in-flight side launch was built for: drain first
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->side) HIPCHK(hipStreamSynchronize(e->side));
    e->opt.*(d->field) = (int)value;
    return ISK_OK;
}

// read-only keys besides the options: what isk_create decided for this shape
extern "C" int isk_get_option(isk_engine* e, const char* key, int64_t* value) {
    if (!e || !key || !value) return fail(ISK_EINVAL, "null argument");
    if (const OptDesc* d = find_option(key)) { *value = e->opt.*(d->field); return ISK_OK; }
    if (!std::strcmp(key, "forward_mode")) { *value = e->cfg.forward_mode; return ISK_OK; }      // the EFFECTIVE arithmetic (3 falls back to 1)
    if (!std::strcmp(key, "layered")) { *value = e->layered ? 1 : 0; return ISK_OK; }
    if (!std::strcmp(key, "planes")) { *value = e->planes ? 1 : 0; return ISK_OK; }
    if (!std::strcmp(key, "nparams_engine")) { *value = e->nparams; return ISK_OK; }
    if (!std::strcmp(key, "dp_peer_active")) { *value = e->last_run_peer ? 1 : 0; return ISK_OK; }           // the last data-parallel run read its peers' gradients directly
#ifdef ISK_EXPERIMENTAL
    if (!std::strcmp(key, "fuse_tail_active")) { *value = e->last_fwdtail ? 1 : 0; return ISK_OK; }                 // the last gradient ran train_fwd1n + tail as one launch
    if (!std::strcmp(key, "train_persist_active")) { *value = e->last_epoch_persist ? 1 : 0; return ISK_OK; }   // the last epoch ran as one persistent launch
#endif
    if (!std::strcmp(key, "wide_fwd")) { *value = e->xh16 ? 1 : 0; return ISK_OK; }              // resident data runs the fused wide forward
    return fail(ISK_EINVAL, std::string("unknown option: ") + key);
}
