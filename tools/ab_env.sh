#!/usr/bin/env bash
# A/B of one tuning knob on the replay bench, interleaved on one box: tools/ab_env.sh VAR "v1 v2" [repeats] [bench args]
# The knobs exist in the -DH2MI_AB library only (make -C halo2-scaffold_amd/csrc ab): it is selected here with H2MI_LIBRARY.
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
VAR=$1; VALS=$2; REP=${3:-2}; shift 3 || true
export H2MI_LIBRARY="$PWD/halo2-scaffold_amd/libh2mi_ab.so"
[ -f "$H2MI_LIBRARY" ] || { echo "build it first: make -C halo2-scaffold_amd/csrc ab" >&2; exit 2; }
for r in $(seq "$REP"); do
  for v in $VALS; do
    env "$VAR=$v" python3 bench.py --no-cpu-baseline --no-create-proof --no-msm-only "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
s=sys.stdin.read(); d=json.loads(s[s.index('{'):])
print('$VAR=$v', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['issue_roofline']['frac'])"
  done
done
