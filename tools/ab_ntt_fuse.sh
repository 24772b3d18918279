#!/usr/bin/env bash
# A/B of the fused first / last NTT rounds inside ONE build (libh2mi_ab.so, -DH2MI_AB: H2MI_NTT_NO_FUSE=1 restores the LDS round trips),
# alternating on one box: tools/ab_ntt_fuse.sh [REPEATS] [sizes...]
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
R=${1:-3}; shift || true
SIZES=${*:-"16 18 20 21 22 24"}
export H2MI_LIBRARY=$(readlink -f halo2-scaffold_amd/libh2mi_ab.so)
for r in $(seq $R); do
  echo "== unfused"; H2MI_NTT_NO_FUSE=1 python3 tools/ntt_sweep.py $SIZES 2>/dev/null | cut -c1-110
  echo "== fused";   python3 tools/ntt_sweep.py $SIZES 2>/dev/null | cut -c1-110
done
