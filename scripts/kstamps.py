"""start times of consecutive training-kernel launches without a profiler (diagnostic build -DISK_KSTAMP):
DEVFLAGS_ALL=-DISK_KSTAMP scripts/devbuild.sh $PWD/devlib/kstamp.so; ISK_LIB=$PWD/devlib/kstamp.so python scripts/kstamps.py"""
import ctypes, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
synth = importlib.import_module("isokann_jl_amd.synth")
c = synth.WORKLOADS["c3"]
xs, ys = synth.synthetic_coords(c["atoms"], c["N"], c["K"], seed=0)
feat = pkg.FeaturesPairs(synth.random_pairs(c["atoms"], c["feat"], seed=0))
iso = pkg.Iso(pkg.SimulationData(xs, ys, feat), model=pkg.densenet(pkg.pairnet_widths(c["feat"], nout=3), layernorm=True),
              opt=pkg.AdamRegularized(), minibatch=int(os.environ.get("KS_MB", "1000")), forward_mode=3, featurize_every_iteration="--cached" not in sys.argv)
pkg.run(iso, 12, 1)
lib = iso._lib
buf = (ctypes.c_ulonglong * 2048)()
n = ctypes.c_uint(0)
lib.isk_debug_kstamps(buf, ctypes.byref(n))
t = np.array(list(buf), dtype=np.uint64)
cnt = int(n.value)
t = t[:min(cnt, 2048)]
ids = (t >> np.uint64(56)).astype(int)
ts = (t & np.uint64((1 << 56) - 1)).astype(np.int64) / 100.0        # us (100 MHz)
names = {1: "fwd1n", 2: "tail", 3: "grad1n", 4: "opt"}
last = ts[-40:]; lid = ids[-40:]
for i in range(len(last)):
    print(f"{i:3d} {names.get(lid[i], lid[i]):7s} start {last[i] - last[0]:8.1f} us   since previous start {0.0 if i == 0 else last[i] - last[i - 1]:6.1f}")
