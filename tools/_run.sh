cd $GRAFT_REPO_ROOT
V=subphaser_amd/lib/variants
bash tools/ab_round.sh "za1|$V/lib_za1.so|SP_MZ_BITS_PER_KEY=4|" "za2|$V/lib_za2.so|SP_MZ_BITS_PER_KEY=4|" "za3|$V/lib_za3.so|SP_MZ_BITS_PER_KEY=4|" "za6|$V/lib_za6.so|SP_MZ_BITS_PER_KEY=4|" "zu4|$V/lib_zu4.so|SP_MZ_BITS_PER_KEY=4|" "zu1|$V/lib_zu1.so|SP_MZ_BITS_PER_KEY=4|" 2>&1 | grep -v "^   " > /dev/null
for n in za1 za2 za3 za6 zu4 zu1; do python - <<P
import json
d=json.loads(open("gpurun_out/ab_$n.json").read().strip().splitlines()[-1])
st=d["stages"]
print("$n", {k:v["ms_per_step"] for k,v in st.items() if k.startswith("k5_map")}, "step", d["ms_per_step"], "mapped", d["config"]["mapped_positions"])
P
done
