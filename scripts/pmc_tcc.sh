#!/bin/bash
# L2 (TCC) read-request split of the dominant forward kernel: scripts/pmc_tcc.sh <tag> [bench.py args...] -> gpurun_out/pmc_tcc_<tag>.json
# One rocprofv3 --pmc pass per counter group (--kernel-trace only).  ISK_LIB=<ablation build> is honoured (bench.py loads it).  RDREQ counts the L2's memory-side read requests, RDREQ_32B the
# 32-byte ones among them (the others are 64 B), HIT / MISS the L2 lookups: (RDREQ - RDREQ_32B) * 64 + RDREQ_32B * 32 is the byte
# count FETCH_SIZE approximates.
export ISK_BENCH_NO_PMC=1     # (bench.py would otherwise run its own rocprofv3 counter passes for roofline.traffic)
TAG=$1; shift
BARGS="$*"
R=$(pwd); P=$R/gpurun_out/pmc_tcc_$TAG; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $P/$n -o $n -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 1 --sustained-s 0 --no-f32-reference --spinup-s 0 $BARGS > /dev/null 2> $P/$n.err || { echo "pass $n failed:"; tail -3 $P/$n.err; }
}
run rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run rdw TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run hit TCC_HIT_sum TCC_MISS_sum
run req TCC_REQ_sum TCC_READ_sum
run wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
cd $R && python3 - "$P" "$TAG" <<'PY'
import collections, csv, glob, json, os, re, sys
root, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(float); names = {}
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = re.sub(r"\(.*$", "", r["Kernel_Name"]).strip()
    for (d, c), v in per.items():
        acc[names[d]][c].append(v)
out = {k: {c: max(x) for c, x in v.items()} for k, v in acc.items() if any(s in k for s in ("mlp_fwd", "wide_fwd", "featurize"))}
for k, v in out.items():
    if "TCC_EA0_RDREQ_sum" in v:
        r32 = v.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        r64, r128 = v.get("TCC_EA0_RDREQ_64B_sum"), v.get("TCC_EA0_RDREQ_128B_sum")
        if r64 is not None and r128 is not None:
            v["read_bytes_from_requests"] = r32 * 32 + r64 * 64 + r128 * 128
        else:
            v["read_bytes_from_requests_if_all_64B"] = (v["TCC_EA0_RDREQ_sum"] - r32) * 64 + r32 * 32
    if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
        v["l2_hit_rate"] = v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
json.dump(out, open("gpurun_out/pmc_tcc_%s.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
