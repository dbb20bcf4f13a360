"""Summarise the rocprofv3 --pmc passes of scripts/pmc.sh into the JSON committed as profiles/rNN_pmc_summary.json."""
import collections, csv, glob, json, os, re, sys

root = sys.argv[1]
bench_args = sys.argv[2] if len(sys.argv) > 2 else ""


def short(name):
    return re.sub(r"\(.*$", "", name).strip()


def per_kernel(passname):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, passname, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
        for (d, c), v in per_dispatch.items():
            acc[names[d]][c].append(v)
    return acc


out = {}
for pn, cn in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    a = per_kernel(pn)
    out[cn] = {k: {"avg_KB": sum(v[cn]) / len(v[cn]), "max_KB": max(v[cn]), "dispatches": len(v[cn])} for k, v in a.items() if cn in v}
sq = per_kernel("sq")
out["SQ"] = {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in sq.items()}
dom = [k for k in out["FETCH_SIZE"] if "mlp_fwd" in k or "wide_fwd" in k]
dom = max(dom, key=lambda k: out["FETCH_SIZE"][k]["avg_KB"]) if dom else None
out["dominant_kernel"] = dom
if dom:
    # gfx950: FETCH_SIZE tallies 64 B per 128-B request of a wide coalesced read -> doubled; WRITE_SIZE exact; units KB
    out["mlp_fwd_traffic_bytes_per_launch"] = (2 * out["FETCH_SIZE"][dom]["max_KB"] + out["WRITE_SIZE"].get(dom, {"max_KB": 0})["max_KB"]) * 1024
    s = out["SQ"].get(dom, {})
    # the dominant kernel's own duration in the SQ pass (kernel trace of the same run): effective clock = GRBM_GUI_ACTIVE / 8 / duration
    durs = []
    for f in glob.glob(os.path.join(root, "sq", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if short(r["Kernel_Name"]) == dom:
                durs.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    if durs and s.get("GRBM_GUI_ACTIVE"):
        out["dominant_kernel_duration_us_in_sq_pass"] = sum(durs) / len(durs) / 1e3
        out["dominant_kernel_effective_clock_GHz"] = s["GRBM_GUI_ACTIVE"] / 8 / (sum(durs) / len(durs))
    if s.get("GRBM_GUI_ACTIVE"):
        # SQ_VALU_MFMA_BUSY_CYCLES sums the 4 SIMDs of every CU in cycles; GRBM_GUI_ACTIVE sums the 8 XCDs
        out["dominant_kernel_derived"] = {
            "mfma_busy_fraction_of_kernel": s["SQ_VALU_MFMA_BUSY_CYCLES"] / (s["GRBM_GUI_ACTIVE"] / 8 * 256 * 4),
            "wave_cycle_shares": {"wait_any": s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"], "wait_inst_any": s["SQ_WAIT_INST_ANY"] / s["SQ_WAVE_CYCLES"],
                                  "active_inst": s["SQ_ACTIVE_INST_ANY"] / s["SQ_WAVE_CYCLES"]}}
out["note"] = ("rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | SQ set) with --kernel-trace only; command: python bench.py --steps 2 "
               "--warmup 1 --no-cpu-baseline --profile-steps 1 " + bench_args + ". gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE tallies 64 B per 128-B "
               "request of a wide coalesced read -> doubled; WRITE_SIZE exact; units KB. SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count "
               "quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles, GRBM_GUI_ACTIVE sums the 8 XCDs.")
json.dump(out, sys.stdout, indent=1)
