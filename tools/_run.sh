cd $GRAFT_REPO_ROOT
for v in ht256 ht768 ht1024 hpf4 hpf16 ht256pf16; do
  export SUBPHASER_HIP_LIB=$PWD/subphaser_amd/lib/variants/lib_$v.so
  echo "== $v"; timeout 300 python tools/k1_bench.py 667e6 2 15 2>&1 | tail -1 | cut -c60-200
done
