"""Shared check of the data-parallel GPU tests: a sharded run against the single-process run on the full data.

The two runs sum the same gradient in a different order.  Nesterov is linear in the gradient: the parameters agree to
rounding, asserted strictly.  Adam's first step moves a parameter by lr * g / (|g| + eps), i.e. by +-lr whatever |g| is:
an entry whose gradient is zero up to f32 rounding can step in opposite directions in the two runs.  Such entries are not
tolerated wholesale -- every out-of-tolerance entry must have a FIRST-STEP gradient at rounding level, shown on the reference
engine (isk_batch_grads at the initial parameters on the first global batch)."""
import numpy as np

ROUNDING_LEVEL = 1e-4          # |g_i| <= 1e-4 * max|g|: below what two f32 summation orders of ~100 O(max|g|) terms can resolve


def assert_params_match(pkg, got, want, opt_kind, make_ref, first_batch, lr, nsteps):
    """got / want: flat parameters of the sharded run and of the single-process reference after `nsteps` optimizer steps;
    make_ref(): a fresh reference engine at the INITIAL parameters; first_batch: sample indices of the first global batch"""
    if opt_kind == "nesterov":
        np.testing.assert_allclose(got, want, atol=5e-6, rtol=5e-4)
        return
    off = np.abs(got - want) > 5e-6 + 5e-4 * np.abs(want)
    if not off.any():
        return
    ref = make_ref()
    try:
        pkg.isotarget(ref)                                            # the device-resident target of the first iteration
        g0, _ = pkg.batch_grads(ref, None, np.asarray(first_batch, np.int32))
    finally:
        ref.close()
    thr = ROUNDING_LEVEL * np.abs(g0).max()
    assert np.all(np.abs(g0[off]) <= thr), (
        f"{off.sum()} of {off.size} parameters differ; the largest first-step |g| among them is {np.abs(g0[off]).max():.3e} "
        f"(rounding level {thr:.3e}): not an Adam sign flip of a rounding-level gradient")
    assert off.mean() <= 2e-3, f"{off.sum()} of {off.size} parameters differ"
    assert np.abs(got - want).max() <= 2 * nsteps * lr + 1e-6
