#!/bin/bash
# Compiles the solver translation units with -Rpass-analysis=kernel-resource-usage and writes the per-kernel table the
# round's profiles/ keeps (zero ScratchSize and zero VGPR spills in every instance of the L1 kernels).
#   tools/solver_resources.sh profiles/r04_solver_resources.txt
set -eu
OUT=$(realpath -m "${1:-/dev/stdout}")
cd "$(dirname "$0")/../phenotypeseeker_amd/csrc"
TMP=$(mktemp -d)
for f in solver solver_l1_gram solver_l1_gg solver_l2; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Rpass-analysis=kernel-resource-usage \
      -c $f.hip -o "$TMP/$f.o" > "$TMP/$f.txt" 2>&1 &
done
wait
python3 - "$TMP" > "$OUT" <<'PY'
import re, subprocess, sys
tmp = sys.argv[1]
print("# hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage, one line per kernel instance (tools/solver_resources.sh)")
print("# %-118s %5s %5s %5s %8s %4s %7s %7s %6s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ", "sgpr-sp", "vgpr-sp", "LDS"))
bad = 0
for f in ("solver", "solver_l1_gram", "solver_l1_gg", "solver_l2"):
    txt = open("%s/%s.txt" % (tmp, f)).read()
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split("\n")[0].split(" [-R")[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0]
        g = lambda k: re.search(re.escape(k) + r": (\S+)", b).group(1)
        row = (g("VGPRs"), g("AGPRs"), g("SGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"), g("SGPRs Spill"), g("VGPRs Spill"), g("LDS Size [bytes/block]"))
        print("%-120s %5s %5s %5s %8s %4s %7s %7s %6s" % ((f + ".hip: " + dem.replace("void ", ""),) + row))
        bad += int(row[3]) != 0 or int(row[6]) != 0
print("# instances with scratch or VGPR spills: %d" % bad)
PY
rm -rf "$TMP"
