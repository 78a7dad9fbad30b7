#!/bin/bash
# Parity campaigns of the production kernels on the GPU box: tools/campaign.sh <tag> [seed]
#   A  production == reference-order kernel on every pixel (ids, t, radiance, ray count): random cameras on the spheres,
#      mesh and instanced scenes, the full 1M-triangle mesh scene (large-scene variant of the path pool), the glass
#      variant, random triangle soups                                   -> gpurun_out/<tag>_campaign_kernels.txt
#   B  production == CPU oracle (16 host threads), path tracing and REF mode
#                                                                       -> gpurun_out/<tag>_campaign_oracle.txt
# Each step under its own time limit; a failing step ends the script.
cd "${GRAFT_REPO_ROOT:-.}"
TAG=${1:-run}; SEED=${2:-2026}
A=gpurun_out/${TAG}_campaign_kernels.txt; B=gpurun_out/${TAG}_campaign_oracle.txt
: > $A; : > $B
run() { out=$1; shift; timeout -k 10 "$@" >> $out 2>&1 || { tail -3 $out; exit 1; }; }
run $A 900 python3 -u tools/parity_campaign.py --seed $SEED --cameras 24 --spp 8 --workloads spheres,mesh,instanced
run $A 600 python3 -u tools/parity_campaign.py --seed $SEED --cameras 12 --spp 8 --workloads mesh --mesh-tris 1000000
run $A 600 python3 -u tools/parity_campaign.py --seed $SEED --cameras 16 --spp 8 --workloads glass
run $A 600 python3 -u tools/parity_campaign.py --seed $SEED --cameras 60 --spp 8 --workloads soup
run $B 900 python3 -u tests/campaign_oracle.py --seed $SEED --cameras 12 --spp 2 --workloads spheres,mesh,instanced,glass,soup
run $B 600 python3 -u tests/campaign_oracle.py --seed $SEED --mode ref --cameras 40 --workloads spheres,mesh,instanced,soup
for f in $A $B; do
  echo "== $f: $(grep -c camera $f) frames"; grep "total mismatches" $f
  python3 - $f <<'PY'
import re, sys
print("segments in all frames:", sum(int(m.group(1)) for m in re.finditer(r"(\d+) segments", open(sys.argv[1]).read())))
PY
done
