#!/bin/bash
# The round's parity campaign (production kernel == reference-order kernel on every pixel): tools/campaign_r02.sh
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/parity_campaign.txt
: > $OUT
timeout -k 10 900 python3 -u tools/parity_campaign.py --cameras 30 --spp 8 --workloads spheres,mesh,instanced >> $OUT 2>&1 || { tail -3 $OUT; exit 1; }
timeout -k 10 600 python3 -u tools/parity_campaign.py --cameras 16 --spp 8 --workloads mesh --mesh-tris 1000000 >> $OUT 2>&1 || { tail -3 $OUT; exit 1; }
timeout -k 10 600 python3 -u tools/parity_campaign.py --cameras 40 --spp 8 --workloads soup >> $OUT 2>&1 || { tail -3 $OUT; exit 1; }
grep -c "camera" $OUT; grep "total mismatches" $OUT
python3 - <<'PY'
import re
n = sum(int(m.group(1)) for m in re.finditer(r"(\d+) segments", open("gpurun_out/parity_campaign.txt").read()))
print("segments in all frames:", n)
PY
