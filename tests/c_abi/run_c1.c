/* Configuration C1 (BASELINE.json configs[0]: D = 2, N = 100, K = 5, one-dimensional chi, default pairnet [2,2,1,1] + LayerNorm,
 * NesterovRegularized, minibatch 100) driven through the C ABI by a plain C program -- the call sequence a Julia host makes with
 * ccall (julia/IsokannHIP.jl), with no ctypes marshalling in between:
 *
 *     isk_config_default -> isk_create -> isk_set_params -> isk_set_data -> isk_target -> isk_run -> isk_chis / isk_koopman
 *                                      (Iso(data; model, opt), src/iso.jl:17-43;  isotarget, src/isotarget.jl:10-12;
 *                                       run!(iso, n), src/iso.jl:72-94;  chis, src/iso.jl:203;  koopman, src/isotarget.jl:20)
 *
 * and compared with the committed golden vectors (tests/golden/c1_shiftscale_nesterov.npz, dumped to c1_golden.h by
 * make_c1_header.py) at the tolerances tests/test_gpu_parity.py::test_run_matches_golden uses.
 *
 *     gcc -std=c99 -Wall -Iinclude tests/c_abi/run_c1.c -Lisokann.jl_amd/lib -lisokann_hip -lm -o run_c1
 *
 * Exit code 0: every comparison inside its bound.  1: a mismatch.  2: an ABI call failed (no GPU: isk_create says so). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "isokann_hip.h"
#include "c1_golden.h"

#define CALL(x) do { int st_ = (x); if (st_ != ISK_OK) { \
    fprintf(stderr, "%s -> %d (%s): %s\n", #x, st_, isk_status_string(st_), isk_last_error()); return 2; } } while (0)

static int bad;
static void cmp(const char* what, const double* want, const double* got, int n, double atol, double rtol) {
    double worst = 0;
    int at = 0, i;
    for (i = 0; i < n; ++i) {
        const double err = fabs(got[i] - want[i]), lim = atol + rtol * fabs(want[i]);
        const double ratio = err == 0 ? 0 : (lim > 0 ? err / lim : HUGE_VAL);       /* NaN compares false below: caught by !(worst <= 1) */
        if (!(ratio <= worst)) { worst = ratio; at = i; }
    }
    printf("%-14s n = %3d  worst error / bound = %.3g at [%d] (want %.9g, got %.9g)  %s\n", what, n, worst, at, want[at], got[at],
           worst <= 1 ? "ok" : "MISMATCH");
    if (!(worst <= 1)) bad = 1;
}

int main(int argc, char** argv) {
    isk_config cfg;
    isk_engine* e = NULL;
    int64_t np = 0;
    float buf[C1_N];
    double got[C1_N], losses[C1_NITER], want1[1], got1[1];
    float params[C1_NPARAMS];
    int i, mode = argc > 1 ? atoi(argv[1]) : 0;

    if (isk_abi_version() != ISK_ABI_VERSION) { fprintf(stderr, "ABI version %d, header %d\n", isk_abi_version(), ISK_ABI_VERSION); return 2; }
    isk_config_default(&cfg);                       /* the reference's defaults: sigmoid, LayerNorm, Nesterov(1e-3, 1e-4), minibatch 100 */
    if (cfg.opt_kind != ISK_OPT_NESTEROV || cfg.minibatch != C1_MINIBATCH || cfg.layernorm != 1) { fprintf(stderr, "defaults differ from src/iso.jl:10,18\n"); return 1; }
    cfg.n_layers = C1_NLAYERS;
    for (i = 0; i <= C1_NLAYERS; ++i) cfg.widths[i] = c1_widths[i];
    cfg.lr = C1_LR; cfg.reg = C1_REG;
    cfg.featurizer = ISK_FEAT_IDENTITY;             /* SimulationData(xs, ys) without a featurizer: docs/src/introduction.md:36-37 */
    cfg.target_kind = ISK_TARGET_SHIFTSCALE;
    cfg.forward_mode = mode;                        /* 0 = Float32 MFMA, 1 = split-bf16, 3 = split-f16 */
    CALL(isk_create(&cfg, &e));
    CALL(isk_nparams(e, &np));
    if (np != C1_NPARAMS) { fprintf(stderr, "nparams %lld, want %d\n", (long long)np, C1_NPARAMS); return 1; }
    CALL(isk_set_params(e, c1_params0, np));
    CALL(isk_set_data(e, c1_xs, c1_ys, C1_D, C1_K, C1_N));

    CALL(isk_target(e, buf));                       /* isotarget(iso): shiftscale(K chi) */
    for (i = 0; i < C1_N; ++i) got[i] = buf[i];
    cmp("target0", c1_target0, got, C1_N, 2e-4, 0);

    CALL(isk_run(e, C1_NITER, 1, c1_perms, losses));   /* run!(iso, 5): injected DataLoader shuffles */
    cmp("losses", c1_losses, losses, C1_NITER, 0, 2e-4);

    CALL(isk_chis(e, buf));
    for (i = 0; i < C1_N; ++i) got[i] = buf[i];
    cmp("chis", c1_chis, got, C1_N, 5e-5, 0);
    CALL(isk_koopman(e, buf));
    for (i = 0; i < C1_N; ++i) got[i] = buf[i];
    cmp("koopman", c1_koopman, got, C1_N, 5e-5, 0);

    CALL(isk_get_params(e, params, np));            /* cpu(iso) / save: the trained parameters in Flux order */
    {
        double pw[C1_NPARAMS], pg[C1_NPARAMS];
        for (i = 0; i < C1_NPARAMS; ++i) { pw[i] = c1_params_final[i]; pg[i] = params[i]; }
        cmp("params", pw, pg, C1_NPARAMS, 2e-5, 1e-4);
    }
    /* the target's known answer (src/isotarget.jl:36-42): min 0, max 1 exactly */
    CALL(isk_target(e, buf));
    want1[0] = 0; got1[0] = buf[0];
    for (i = 1; i < C1_N; ++i) if (buf[i] < got1[0]) got1[0] = buf[i];
    cmp("min(target)", want1, got1, 1, 0, 0);
    want1[0] = 1; got1[0] = buf[0];
    for (i = 1; i < C1_N; ++i) if (buf[i] > got1[0]) got1[0] = buf[i];
    cmp("max(target)", want1, got1, 1, 0, 0);
    isk_destroy(e);
    printf(bad ? "C1 through the C ABI: MISMATCH\n" : "C1 through the C ABI: ok (forward mode %d)\n", mode);
    return bad;
}
