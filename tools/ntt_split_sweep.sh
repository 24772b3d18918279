#!/bin/bash
# pass-split sweep of the NTT (H2MI_NTT_SPLIT): which decomposition of log_n into <= 3 LDS passes is fastest
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
export H2MI_LIBRARY="$PWD/halo2-scaffold_amd/libh2mi_ab.so"  # H2MI_NTT_SPLIT exists in the -DH2MI_AB build only
run() { H2MI_NTT_SPLIT=$2 python3 tools/ntt_sweep.py $1 2>/dev/null | cut -c1-105 | sed "s/\$/ split=$2/"; }
for sp in 10,10 8,8,4 8,6,6 7,7,6 9,7,4 9,9,2 8,7,5 10,6,4; do run 20 $sp; done
for sp in 7,7,7 8,8,5 8,7,6 9,8,4 10,7,4 9,6,6 10,10,1; do run 21 $sp; done
for sp in 8,7,7 8,8,6 9,9,4 10,8,4 10,6,6; do run 22 $sp; done
for sp in 8,8,8 9,9,6 10,10,4 10,8,6 9,8,7; do run 24 $sp; done
