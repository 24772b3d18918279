#!/bin/bash
# window-width sweep of the MSM at 2^20 .. 2^22 (uniform scalars): accumulation, partition and bucket-reduction times per c.
# Runs on the GPU box; output -> profiles/rNN_msm_sweep.txt (c = 18 .. 20 measured, not argued away).
cd "$GRAFT_REPO_ROOT"
for k in ${KS:-20 21 22}; do
  for c in ${CS:-15 16 17 18 19 20}; do
    H2MI_MSM_C=$c python3 tools/msm_sweep.py $k 2>/dev/null | grep "^k="
  done
done
