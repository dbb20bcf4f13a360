"""phase timeline of train_tail (diagnostic build -DISK_TAIL_STAMP): ISK_LIB=.../tstamp.so python scripts/tail_stamps.py [bench args]"""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
synth = importlib.import_module("isokann_jl_amd.synth")
cached = "--cached" in sys.argv
c = synth.WORKLOADS["c3"]
xs, ys = synth.synthetic_coords(c["atoms"], c["N"], c["K"], seed=0)
feat = pkg.FeaturesPairs(synth.random_pairs(c["atoms"], c["feat"], seed=0))
iso = pkg.Iso(pkg.SimulationData(xs, ys, feat), model=pkg.densenet(pkg.pairnet_widths(c["feat"], nout=3), layernorm=True),
              opt=pkg.AdamRegularized(), minibatch=1000, forward_mode=3, featurize_every_iteration=not cached)
for _ in range(5):
    pkg.run(iso, 1, 1)
lib = iso._lib
buf = (ctypes.c_ulonglong * 32)()
lib.isk_debug_tail_stamps(buf)
t = np.array(list(buf), dtype=np.int64)
names = ["prologue loads + LDS staging", "forward layers", "loss + reduce", "backward", "delta1 store"]
for i, n in enumerate(names):
    print(f"{n:32s} {(t[i + 1] - t[i]) / 100.0:7.2f} us")
g = t[8:13]
if g[4] > g[0] > 0:
    for i, n in enumerate(["grad1n: partial-vector reduction", "grad1n: main loop", "grad1n: LDS combine + D", "grad1n: epilogue"]):
        print(f"{n:36s} {(g[i + 1] - g[i]) / 100.0:7.2f} us")
print(f"{'total':32s} {(t[5] - t[0]) / 100.0:7.2f} us   shader clock {(t[21] - t[16]) / ((t[5] - t[0]) / 100.0) / 1e3:.2f} GHz")
