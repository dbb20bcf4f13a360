"""side-stream featurizer alone (run!(iso, n, 0): no training beside it): python scripts/time_feat.py"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
synth = importlib.import_module("isokann_jl_amd.synth")
c = synth.WORKLOADS["c3"]
xs, ys = synth.synthetic_coords(c["atoms"], c["N"], c["K"], seed=0)
feat = pkg.FeaturesPairs(synth.random_pairs(c["atoms"], c["feat"], seed=0))
iso = pkg.Iso(pkg.SimulationData(xs, ys, feat), model=pkg.densenet(pkg.pairnet_widths(c["feat"], nout=3), layernorm=True),
              opt=pkg.AdamRegularized(), minibatch=1000, forward_mode=3, featurize_every_iteration=True)
pkg.run(iso, 3, 1)
iso.timers(enable=True, reset=True)
pkg.run(iso, 10, 1)
t = iso.timers(enable=False)
print("with training beside:", {k: round(v[0] / 10, 3) for k, v in t.items() if v[1]})
# ISK_SIDE_MIN_STEPS=0: the throttled side-stream launch even when no training phase follows (epochs = 0): the featurizer ALONE
iso.timers(enable=True, reset=True)
pkg.run(iso, 10, 0)
t = iso.timers(enable=False)
print("no training (epochs = 0):", {k: round(v[0] / 10, 3) for k, v in t.items() if v[1]})
