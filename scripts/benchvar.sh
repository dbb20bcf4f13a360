for w in 1 2; do
  ISK_FWD_WAVES_PER_SIMD=$w timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/bv_$w.json
  python - <<PY
import json
d=json.load(open("gpurun_out/bv_$w.json"))
print("wps", $w, round(d["ms_per_step"],3), round(d["value"]/1e6,2), round(d["roofline"]["frac"],3), {k:(round(v,3) if v else v) for k,v in d["roofline"]["phases_ms"].items()})
PY
done
