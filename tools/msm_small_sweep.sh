#!/usr/bin/env bash
# defaults through the product library, then window width / cells per lane through the -DH2MI_AB build; run on the GPU box:
#   bash tools/msm_small_sweep.sh > gpurun_out/msm_small_sweep.txt
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
python3 tools/msm_small_sweep.py 5 8 10 12 13 14 15 16 17
AB="$PWD/halo2-scaffold_amd/libh2mi_ab.so"
# the latency path pinned (no switch to the general pipeline under a deep queue): the third column of the path comparison
H2MI_LIBRARY="$AB" H2MI_MSM_NO_AUTO_STREAM=1 python3 tools/msm_small_sweep.py 12 13 14 15 | grep "small path"
for c in 5 6 7; do
  H2MI_LIBRARY="$AB" H2MI_MSM_SMALL_C=$c python3 tools/msm_small_sweep.py 8 12 14 16 | grep "small path"
done
