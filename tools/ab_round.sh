#!/bin/bash
# Experiment helper (GPU box): run bench.py once per "name|lib|ENV=.. ENV=..|bench args" spec and print the stage table.
#   tools/ab_round.sh "base|subphaser_amd/lib/variants/lib_base.so||" "new|||" ...
# An empty lib = the shipped library.  Lines land in gpurun_out/ab_<name>.json.
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for spec in "$@"; do
    IFS='|' read -r name lib envs args <<< "$spec"
    (
        [ -n "$lib" ] && export SUBPHASER_HIP_LIB=$PWD/$lib
        for e in $envs; do export "$e"; done
        timeout 900 python bench.py --no-cpu-baseline --steps 5 --warmup 2 $args 2> gpurun_out/ab_$name.err | tail -1 > gpurun_out/ab_$name.json
    )
    python tools/stage_ms.py gpurun_out/ab_$name.json
done
