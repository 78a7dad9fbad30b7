#!/bin/bash
# A second, larger campaign with other cameras, scenes and sample counts: tools/campaign_r02b.sh
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/parity_campaign_b.txt
: > $OUT
timeout -k 10 1000 python3 -u tools/parity_campaign.py --seed 31337 --cameras 40 --spp 16 --workloads spheres,mesh,instanced >> $OUT 2>&1 || { tail -3 $OUT; exit 1; }
timeout -k 10 900 python3 -u tools/parity_campaign.py --seed 31337 --cameras 24 --spp 16 --workloads mesh --mesh-tris 1000000 >> $OUT 2>&1 || { tail -3 $OUT; exit 1; }
timeout -k 10 900 python3 -u tools/parity_campaign.py --seed 31337 --cameras 120 --spp 8 --workloads soup >> $OUT 2>&1 || { tail -3 $OUT; exit 1; }
grep -c "camera" $OUT; grep "total mismatches" $OUT
python3 - <<'PY'
import re
n = sum(int(m.group(1)) for m in re.finditer(r"(\d+) segments", open("gpurun_out/parity_campaign_b.txt").read()))
print("segments in all frames:", n)
PY
