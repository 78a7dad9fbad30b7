#!/bin/bash
# REF-mode frame time against the regrouping thresholds of kFast: tools/sweep_ref.sh "<refillBelow list>" "<nodeMin list>"
cd "${GRAFT_REPO_ROOT:-.}"
for rb in $1; do
  for nm in $2; do
    RTAMD_REFILL_BELOW=$rb RTAMD_NODE_MIN=$nm timeout -k 10 120 python3 tools/ref_frame_latency.py 100 2>/dev/null | sed "s/^/refillBelow $rb nodeMin $nm: /"
  done
done
