#!/bin/bash
# Campaign of the dielectric variant (instanced scene, every third instance glass): tools/campaign_r02c.sh
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/parity_campaign_c.txt
: > $OUT
timeout -k 10 1000 python3 -u tools/parity_campaign.py --seed 4711 --cameras 40 --spp 8 --workloads glass >> $OUT 2>&1 || { tail -3 $OUT; exit 1; }
grep -c "camera" $OUT; grep "total mismatches" $OUT
python3 - <<'PY'
import re
n = sum(int(m.group(1)) for m in re.finditer(r"(\d+) segments", open("gpurun_out/parity_campaign_c.txt").read()))
print("segments in all frames:", n)
PY
