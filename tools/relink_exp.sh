#!/bin/bash
# Re-links every exp/libvhp_<NAME>.so from its own pool / latency-sweep objects (csrc/build_<NAME>) and the in-tree C-ABI objects
# (csrc/build), after the C ABI changed: seconds instead of a diagnostic rebuild.   usage: tools/relink_exp.sh [NAME ...]
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/visibility-heuristic-path-planner_amd/csrc
names="$@"
[ -z "$names" ] && names=$(ls -d $C/build_* | sed 's/.*build_//')
for n in $names; do
  [ -f $C/build_$n/vhp_pool.o ] || continue
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/exp/libvhp_$n.so $C/build/vhp_capi.o $C/build/vhp_multi.o $C/build_$n/vhp_pool.o $C/build_$n/vhp_lat.o && echo relinked exp/libvhp_$n.so
done
