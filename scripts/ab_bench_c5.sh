#!/bin/bash
# A/B of two builds on ONE box in ONE gpurun call (boxes of the pool differ by 3-7 %): the in-tree library against
# build_abl/libisk_head.so, alternating, three rounds of the C5-shard bench on its bf16 path.
# Make the reference library first, in the build container (objects of the last scripts/devbuild.sh, the changed translation units
# recompiled from `git archive HEAD`):
#   rm -rf /tmp/headtree && mkdir -p /tmp/headtree build_abl && git archive HEAD isokann.jl_amd/csrc include | tar -x -C /tmp/headtree
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c /tmp/headtree/isokann.jl_amd/csrc/<unit>.hip -o /tmp/obj/<unit>_head.o
#   hipcc --offload-arch=gfx950 -fPIC -shared $(sed "s#<unit's object>#/tmp/obj/<unit>_head.o#" /tmp/obj/last_objs.txt) -o build_abl/libisk_head.so
export ISK_BENCH_NO_PMC=1     # (bench.py would otherwise run its own rocprofv3 counter passes for roofline.traffic)
for i in 1 2 3; do
for lib in "" ${ABLIBS:-build_abl/libisk_head.so}; do
ISK_LIB=$lib python bench.py --workload c5shard --forward-mode 2 --steps 10 --warmup 3 --no-cpu-baseline --train-bf16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib' or 'product', round(d['ms_per_step'],3), round(d['roofline']['training_chain']['us_per_step'],1))
"
done; done
