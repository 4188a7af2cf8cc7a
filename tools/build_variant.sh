#!/bin/bash
# Experiment helper: build libsubphaser_hip variants that differ in the -D flags of ONE translation unit.
#   tools/build_variant.sh <name> <unit> "<-D flags>"   ->  subphaser_amd/lib/variants/lib_<name>.so
# (run the bench against it with SUBPHASER_HIP_LIB=...; the standard objects must be built already)
set -e
name=$1; unit=$2; flags=$3
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/subphaser_amd/csrc; obj=$root/subphaser_amd/lib/obj; out=$root/subphaser_amd/lib/variants
mkdir -p $out/obj_$name
/opt/rocm/bin/hipcc $flags -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -Wno-unused-function -Wno-unused-value -Wno-unused-result -c -o $out/obj_$name/$unit.o $src/$unit.hip
others=$(ls $obj/*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/lib_$name.so $out/obj_$name/$unit.o $others
rm -rf $out/obj_$name
echo built $out/lib_$name.so
