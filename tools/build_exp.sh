#!/bin/bash
# Diagnostic builds of the library: tools/build_exp.sh <NAME> "<extra compiler flags>"  -> exp/libvhp_<NAME>.so
# Only the persistent batch kernels (vhp_pool.hip, vhp_lat.hip) are rebuilt with the flags; the C ABI object is
# the in-tree one (csrc/build/vhp_capi.o), so a build takes seconds.  Used with tools/ab_libs.py.
set -e
name=$1; flags=$2
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/visibility-heuristic-path-planner_amd/csrc
mkdir -p $R/exp $C/build_$name
make -s -C $C build/vhp_capi.o build/vhp_multi.o
F="-std=c++17 -O3 -ffp-contract=off -fPIC --offload-arch=gfx950 -I$R/include -I$C -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result $flags"
/opt/rocm/bin/hipcc $F -c -o $C/build_$name/vhp_pool.o $C/vhp_pool.hip &
/opt/rocm/bin/hipcc $F -c -o $C/build_$name/vhp_lat.o $C/vhp_lat.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/exp/libvhp_$name.so $C/build/vhp_capi.o $C/build/vhp_multi.o $C/build_$name/vhp_pool.o $C/build_$name/vhp_lat.o
echo built exp/libvhp_$name.so
