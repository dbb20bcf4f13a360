"""chi / loss error of the C3 sub-sample runs (tests/golden/c3sub_reference.npz) per forward arithmetic, as a markdown table:
python scripts/c3sub_table.py > profiles/rNN_c3sub_error_table.md   (on the GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import io
import contextlib

import __graft_entry__ as ge

ge.build()
pkg = ge.load_package()
from tests import test_gpu_benchshapes as T      # noqa: E402

MODES = {0: "f32 MFMA", 1: "split-bf16 (x3)", 3: "split-f16 (x2), bench default"}
CASES = ["full_adam_1d", "full_adam", "full3_adam", "mb1000_adam", "mb100_nesterov"]
print("| case | featurize | forward arithmetic | max abs chi error | CPU float32 floor (chi) | max rel loss error | float32 floor (loss) | kappa |")
print("|---|---|---|---|---|---|---|---|")
for name in CASES:
    for every in (False, True):
        for mode, label in MODES.items():
            os.environ["ISK_FWD_MODE"] = str(mode)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                lerr, cerr, fl, fc = T._c3sub_run(pkg, name, every)
            kappa = buf.getvalue().strip().split("kappa")[-1].strip()
            print(f"| {name} | {'every iteration' if every else 'cached'} | {label} | {cerr:.2e} | {fc:.2e} | {lerr:.2e} | {fl:.2e} | {kappa} |")
