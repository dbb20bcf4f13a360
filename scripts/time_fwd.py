"""time the forward kernel alone (C3, cached features): python scripts/time_fwd.py [forward_mode]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
synth = importlib.import_module("isokann_jl_amd.synth")
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
c = dict(synth.WORKLOADS["c3"])
if os.environ.get("TF_N"):
    c["N"] = int(os.environ["TF_N"])
xs, ys = synth.synthetic_coords(c["atoms"], c["N"], c["K"], seed=0)
feat = pkg.FeaturesPairs(synth.random_pairs(c["atoms"], c["feat"], seed=0))
iso = pkg.Iso(pkg.SimulationData(xs, ys, feat), model=pkg.densenet(pkg.pairnet_widths(c["feat"], nout=3), layernorm=True),
              opt=pkg.AdamRegularized(), minibatch=1000, forward_mode=mode)
for _ in range(3):
    pkg.koopman(iso)
iso.timers(enable=True, reset=True)
for _ in range(10):
    pkg.koopman(iso)
t = iso.timers(enable=False)
ms, n = t["fwd_ys"]
print(os.environ.get("ISK_LIB", "default").split("/")[-1], "mode", mode, "N", c["N"], "rows", c["N"] * (c["K"] + 1), "fwd ms", round(ms / n, 4))
