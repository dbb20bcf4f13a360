for i in 1 2 3; do
for lib in "" build_abl/libisk_head.so; do
ISK_LIB=$lib python bench.py --workload c5shard --forward-mode 2 --steps 10 --warmup 3 --no-cpu-baseline --train-bf16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib' or 'product', round(d['ms_per_step'],3), round(d['roofline']['training_chain']['us_per_step'],1))
"
done; done
