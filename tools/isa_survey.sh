#!/bin/bash
# Static survey of the compiled kernels (no GPU needed): per kernel the register counts, spills and how much of the machine code
# is the compiler moving spilled SCALAR registers in and out of VGPR lanes (v_readlane / v_writelane) -- the pathology that made
# round 5's k3_eval issue-bound (profiles/r06_notes.md section 6): a kernel whose uniform, loop-invariant conditions are hoisted
# out of a loop as 64-bit lane masks runs out of SGPRs and pays two v_readlane per test at every use.
#   tools/isa_survey.sh [unit ...]     (default: every unit of the library)   ->  one table on stdout
cd "$(dirname "$0")/../subphaser_amd/csrc" || exit 1
UNITS=${@:-sp_count sp_count2 sp_filter sp_map sp_sparse_all sp_enrich sp_fasta sp_text sp_synth sp_ctx}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -Wno-unused-value -Wno-unused-result -Wno-unused-function"
T=$(mktemp -d)
printf "%-34s %5s %5s %7s %7s %7s %9s %9s\n" kernel vgpr sgpr scratch instr valu "rd/wrlane" "behind-1st-loop-header"
for u in $UNITS; do
  /opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only -o $T/$u.s $u.hip 2> /dev/null || { echo "$u: compile failed"; continue; }
  python3 - "$T/$u.s" <<'PY'
import re, sys
txt = open(sys.argv[1]).read().split("\n")
# kernels: from "<name>:" preceded by .globl/.type to ".Lfunc_end"
i = 0
name = None
body = []
meta = {}
def demangle(n):
    m = re.match(r"_Z(\d+)", n)
    if not m: return n
    l = int(m.group(1)); s = m.end()
    base = n[s:s + l]
    rest = n[s + l:]
    t = re.match(r"I((?:L[a-z]\d+E|[a-z])+)E", rest)
    if t:
        args = re.findall(r"L[a-z](\d+)E|([a-z])", t.group(1))
        base += "<" + ",".join(a or {"j": "u32", "y": "u64", "t": "u16"}.get(b, b) for a, b in args) + ">"
    return base
funcs = {}
for ln in txt:
    m = re.match(r"^(_Z\w+|\w+):\s*;?\s*@", ln)
    if m and name is None:
        name = m.group(1); body = []
        continue
    if name is not None:
        if ln.startswith(".Lfunc_end"):
            funcs[name] = body; name = None
        else:
            body.append(ln)
# resource metadata from the .amdhsa block
cur = None
res = {}
for ln in txt:
    m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", ln)
    if m: cur = m.group(1); res[cur] = {}
    elif cur:
        for key in ("next_free_vgpr", "next_free_sgpr", "private_segment_fixed_size"):
            m2 = re.match(r"\s*\.amdhsa_%s\s+(\d+)" % key, ln)
            if m2: res[cur][key] = int(m2.group(1))
        if ".end_amdhsa_kernel" in ln: cur = None
for k, b in funcs.items():
    if k not in res: continue
    instr = [l for l in b if re.match(r"^\s+[a-z]", l) and not l.strip().startswith(".")]
    valu = [l for l in instr if re.match(r"^\s+v_", l)]
    rl = [l for l in instr if re.match(r"^\s+v_(readlane|writelane)", l)]
    # inside loops: between the first loop header comment and the end (a coarse bound), counted per innermost marker
    inloop = 0; seen = False      # (coarse: everything behind the kernel's first loop header, i.e. not the prologue)
    for l in b:
        if "Loop Header" in l or "Inner Loop Header" in l: seen = True
        if seen and re.match(r"^\s+v_(readlane|writelane)", l): inloop += 1
    r = res[k]
    print("%-34s %5d %5d %7d %7d %7d %9d %9d" % (demangle(k)[:34], r.get("next_free_vgpr", 0), r.get("next_free_sgpr", 0),
          r.get("private_segment_fixed_size", 0), len(instr), len(valu), len(rl), inloop))
PY
done
rm -rf $T
