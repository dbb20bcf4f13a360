# usage: bash scripts/benchvar.sh "ENV=1 ENV2=2" "args"  -> one summary line
export ISK_BENCH_NO_PMC=1     # (bench.py would otherwise run its own rocprofv3 counter passes for roofline.traffic)
run() {
  env $1 timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $2 2>/dev/null > gpurun_out/bv.json
  python - "$1 $2" <<'PY'
import json,sys
d=json.load(open("gpurun_out/bv.json"))
print(sys.argv[1], "|", round(d["ms_per_step"],3), "ms", round(d["value"]/1e6,1), "M/s frac", round(d["roofline"]["frac"],3), {k:(round(v,3) if v else v) for k,v in d["roofline"]["phases_ms"].items()})
PY
}
