#!/bin/bash
# incremental developer build: scripts/devbuild.sh [out.so] [extra flags for kernels_fwd_bf16.hip]
# objects are cached in /tmp/obj keyed by source hash + flags; __graft_entry__.build() stays the one-shot build
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/isokann.jl_amd/lib/libisokann_hip.so}; shift || true
EXTRA="$*"
mkdir -p /tmp/obj
OBJS=""
pids=""
for s in $(cd $ROOT/isokann.jl_amd/csrc && ls *.hip | sed 's/\.hip$//'); do
  fl="$DEVFLAGS_ALL"; [ "$s" = kernels_fwd_bf16 ] && fl="$DEVFLAGS_ALL $EXTRA"   # DEVFLAGS_ALL: flags for every translation unit
  h=$(cat $ROOT/isokann.jl_amd/csrc/$s.hip $ROOT/isokann.jl_amd/csrc/*.h $ROOT/isokann.jl_amd/csrc/*.hpp $ROOT/include/isokann_hip.h | sha1sum | cut -c1-12)
  o=/tmp/obj/$s-$h-$(echo "$fl" | sha1sum | cut -c1-6).o
  if [ ! -f $o ]; then
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $fl $ROOT/isokann.jl_amd/csrc/$s.hip -o $o.tmp && mv $o.tmp $o ) &
    pids="$pids $!"
  fi
  OBJS="$OBJS $o"
done
for p in $pids; do wait $p; done
echo $OBJS > /tmp/obj/last_objs.txt          # (ablation builds relink this list with one object replaced)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $OBJS -o $OUT
if [ "$OUT" = "$ROOT/isokann.jl_amd/lib/libisokann_hip.so" ]; then
  python - <<PY
import sys; sys.path.insert(0, "$ROOT")
import __graft_entry__ as g
open(g.LIB + ".srchash", "w").write(g._src_hash())
PY
fi
echo built $OUT
