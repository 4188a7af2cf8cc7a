cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for g in ara peanut wheat; do
  timeout 900 python tools/e2e_cli.py $g /tmp/sp_e2e_$g > gpurun_out/r04_e2e_cli_$g.log 2>&1
  tail -3 gpurun_out/r04_e2e_cli_$g.log | cut -c1-200
done
timeout 900 python tools/feat_bench.py wheat 2000000 > gpurun_out/r04_feature_mode_2M.log 2>&1
tail -3 gpurun_out/r04_feature_mode_2M.log | cut -c1-300
