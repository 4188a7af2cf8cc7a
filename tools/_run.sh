cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q -k "sparse or k17 or medium or feature_join or baseline_shapes" 2>&1 | tail -3
V=subphaser_amd/lib/variants
bash tools/ab_round.sh "k17prev|$V/lib_prev.so||-k 17" "k17new|||-k 17" "k21prev|$V/lib_prev.so||-k 21" "k21new|||-k 21"
