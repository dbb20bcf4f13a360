"""Forward-only timing of the C5-shard Koopman forward (wide network, forward_mode 2) -- also under ablation builds of the wide
kernel whose RESULTS are garbage (ISK_LIB=build_abl/...): python scripts/time_fwd_wide.py [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
from isokann_jl_amd import synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
c = synth.WORKLOADS["c5shard"]
xs, ys = synth.synthetic_coords(c["atoms"], c["N"], c["K"], seed=0)
feat = pkg.FeaturesPairs(synth.random_pairs(c["atoms"], c["feat"], seed=0))
model = pkg.densenet(c["widths"], layernorm=False, seed=0)
iso = pkg.Iso(pkg.SimulationData(xs, ys, feat), opt=pkg.AdamRegularized(), model=model, minibatch=1000, forward_mode=2)
pkg.koopman(iso)
iso.timers(enable=True, reset=True)
for _ in range(reps):
    pkg.koopman(iso)
tm = iso.timers(enable=False)
ms, cnt = tm["fwd_ys"]
print(f"{os.environ.get('ISK_LIB', 'product'):40s} fwd_ys {ms / cnt:.3f} ms over {cnt} launches")
iso.close()
