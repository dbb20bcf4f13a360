"""GPU: randomized parity sweep (also run by tests/test_gpu_fuzz.py) of the C ABI against the fp64 oracle -- shapes the parametrized tests do not list (odd N / K / F,
tiny and ragged batches, 1 .. 10 outputs, both targets and optimizers, every forward mode, both featurize modes, identity / all-pairs /
pair-list featurizers).  One run!() iteration + chis + koopman per case, tolerances of tests/test_gpu_parity.py.
    python scripts/fuzz_parity.py [ncases] [seed]"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from oracle import isokann_oracle as O  # noqa: E402

pkg = ge.load_package()


def sweep(ncases, seed, verbose=True):
  rng = np.random.default_rng(seed)
  big = bool(os.environ.get("FUZZ_BIG"))          # larger shapes: hidden widths to 400 (beyond 160 the layered path takes over), N to 3000
  fails = 0
  skipped = 0
  failed = []
  for case in range(ncases):
      A = int(rng.integers(3, 70 if big else 40))
      kind = rng.choice(["identity", "all", "pairs"])
      N = int(rng.integers(5, 3000 if big else 400))
      K = int(rng.integers(1, 9))
      nout = int(rng.choice([1, 1, 2, 3, 4, 5, 8, 10]))
      mode = int(rng.choice([0, 1, 3]))
      every = bool(rng.integers(0, 2))
      optk = rng.choice(["adam", "nesterov"])
      ln = bool(rng.integers(0, 4) > 0)
      layered = bool(rng.integers(0, 4) == 0)          # the wide-network path forced onto a small network
      if kind == "identity":
          F = int(rng.integers(1, 70))
          if ln and F < 4:
              F = 4           # (LayerNorm over two or three features divides by their tiny spread: Float32 cancellation, not arithmetic)
          xs, ys = rng.normal(size=(F, N)), rng.normal(size=(F, K, N))
          feat, ofeat = pkg.identity, O.identity
      else:
          xs, ys = O.synthetic_coords(A, N, K, seed=int(rng.integers(0, 1 << 30)))
          if kind == "all":
              F = A * (A - 1) // 2
              feat, ofeat = pkg.FeaturesAll(), O.FeaturesAll()
          else:
              F = int(rng.integers(4 if ln else 1, min(900 if big else 300, A * (A - 1) // 2) + 1)) if A * (A - 1) // 2 >= 4 else A * (A - 1) // 2
              pairs = O.random_pairs(A, F, seed=int(rng.integers(0, 1 << 30)))
              feat, ofeat = pkg.FeaturesPairs(pairs), O.FeaturesPairs(pairs)
      if ln and F < 4:
          ln = False        # (LayerNorm over two or three features divides by their tiny spread: Float32 cancellation, not arithmetic)
      if nout > 1 and N < 2 * nout:
          N = 2 * nout + int(rng.integers(0, 50))     # (an ISA simplex needs more points than vertices)
          if kind == "identity":
              xs, ys = rng.normal(size=(F, N)), rng.normal(size=(F, K, N))
          else:
              xs, ys = O.synthetic_coords(A, N, K, seed=int(rng.integers(0, 1 << 30)))
      nl = int(rng.integers(1, 5))
      widths = [F] + [int(rng.integers(max(nout, 1), 400 if big else 60)) for _ in range(nl - 1)] + [nout]
      mb = int(rng.choice([0, 0, 7, 16, 33, 100, 1000]))
      desc = dict(case=case, kind=kind, A=A, F=F, N=N, K=K, widths=widths, mode=mode, every=every, opt=optk, ln=ln, mb=mb, layered=layered)
      try:
          om = O.glorot_model(widths, rng, layernorm=ln)
          for b in om.b:
              b[...] = 0.1 * rng.normal(size=b.shape)
          if ln:
              om.ln_scale[...] = 1 + 0.1 * rng.normal(size=F)
              om.ln_bias[...] = 0.1 * rng.normal(size=F)
          m = pkg.densenet(widths, layernorm=ln)
          m.params = om.get_flat().astype(np.float32)
          popt = pkg.AdamRegularized(1e-3, 1e-4) if optk == "adam" else pkg.NesterovRegularized(1e-3, 1e-4)
          oopt = O.AdamRegularized(1e-3, 1e-4) if optk == "adam" else O.NesterovRegularized(1e-3, 1e-4)
          x64, y64 = np.asarray(xs, np.float32).astype(np.float64), np.asarray(ys, np.float32).astype(np.float64)
          odata = O.SimulationData(x64, y64, ofeat)
          iso = pkg.Iso(pkg.SimulationData(xs, ys, feat), model=m, opt=popt, minibatch=mb, featurize_every_iteration=every, forward_mode=mode, force_layered=layered)
          oiso = O.Iso(odata, om, O.setup(oopt, om), minibatch=mb)
          chi0, ks0 = pkg.chis(iso), pkg.koopman(iso)
          assert np.abs(chi0 - O.chis(oiso)).max() <= 3e-5, ("chi0", np.abs(chi0 - O.chis(oiso)).max())
          assert np.abs(ks0 - O.koopman(oiso)).max() <= 3e-5, ("ks0", np.abs(ks0 - O.koopman(oiso)).max())
          perm = rng.permutation(N)
          try:
              O.run(oiso, 1, 1, perms=[perm])
          except O.DomainError as ex:
              # an EXACTLY singular simplex in Float64 (e.g. one feature, five outputs) is only nearly singular behind a Float32 forward:
              # like Julia's inv() on the reference's Float32 K chi, the engine need not raise -- nothing to compare either way
              try:
                  pkg.run(iso, 1, 1, perms=perm[None, :])
              except pkg.DomainError:
                  pass
              iso.close()
              skipped = skipped + 1
              continue
          # teacher-forced epoch (an ill-conditioned ISA target amplifies the forward's 2e-7 beyond any fixed bound): the ENGINE's
          # target, the oracle's train_batch! on it
          om2 = O.glorot_model(widths, np.random.default_rng(0), layernorm=ln)
          om2.set_flat(om.get_flat() * 0 + m.params.astype(np.float64))
          t = pkg.isotarget(iso)
          if np.abs(t).max() > 20:
              # a simplex this badly conditioned (targets far outside [0, 1]) makes the epoch DIVERGE at lr 1e-3 -- parameters of 1e11 and
              # "model collapsed" within nine steps, in the oracle as in the engine: nothing to compare (seen for one-layer networks)
              iso.close()
              skipped += 1
              continue
          st2 = O.setup(oopt, om2)
          th0 = om2.get_flat().copy()
          try:
              lo = O.train_batch(om2, odata.features[0], t.astype(np.float64), st2, mb, perm=perm)
          except O.DomainError:
              lo = float("inf")
          if not np.isfinite(lo) or np.abs(om2.get_flat() - th0).max() > 0.25:
              # the epoch DIVERGES in the oracle (un-normalised features of magnitude 2 x 465 of them, or a one-layer model, at lr 1e-3
              # with Nesterov: the step exceeds 2 / curvature): parameters of 1e3 .. 1e11 -- chaotic, nothing to compare
              try:
                  pkg.train_batch(iso, None, perm=perm)
              except pkg.DomainError:
                  pass
              iso.close()
              skipped += 1
              continue
          l = pkg.train_batch(iso, None, perm=perm)
          assert abs(l - lo) <= 2e-4 * max(abs(lo), 1e-12), ("loss", l, lo)
          if optk == "nesterov":
              d = np.abs(iso.model.get_params() - om2.get_flat()).max()
              assert d <= 5e-5, ("params", d)
          else:       # Adam turns a rounding-level gradient into a full +-lr step: compare what the parameters DO, not the parameters
              d = np.abs(pkg.chis(iso) - om2(odata.features[0])).max()
              assert d <= 2e-3, ("chi after the epoch", d)
          iso.close()
      except Exception as ex:      # noqa: BLE001
          fails += 1
          failed.append((desc, repr(ex)[:300]))
          print("FAIL", desc, repr(ex)[:300], flush=True)
          traceback.print_exc(limit=2)
  if verbose:
    print(f"{ncases} cases, {skipped} skipped (singular / divergent in the oracle), {fails} failures", flush=True)
  return fails, skipped, failed


if __name__ == "__main__":
    f, _, _ = sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    sys.exit(1 if f else 0)
