#!/usr/bin/env bash
# A/B of two builds of libh2mi.so on ONE box, interleaved (box-to-box variation is +-4 %).  Neither build is ever copied over
# the product library: Python tools select a build with H2MI_LIBRARY=<path>, the C++ examples (RUNPATH) with LD_LIBRARY_PATH
# pointing at a private directory that holds the build under the name libh2mi.so.
#   usage: gpurun -- bash tools/ab_lib.sh <mode> [old.so] [new.so] [repeats]
#   modes: msm (MSM 2^20 + replay step) | proof (2^20 proofs, both hosts) | small (C++ proofs at 2^5 / 2^8 / 2^16 / 2^20) |
#          range (C++ range DEGREE 22 + poseidon) | poly (opening-argument kernels) | evalh (evaluate_h)
#   defaults: old = halo2-scaffold_amd/libh2mi.so.prev (git-ignored, travels with gpurun), new = halo2-scaffold_amd/libh2mi.so
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
MODE=${1:?mode}
OLD=$(readlink -f "${2:-halo2-scaffold_amd/libh2mi.so.prev}")
NEW=$(readlink -f "${3:-halo2-scaffold_amd/libh2mi.so}")
REP=${4:-3}
[ -f "$OLD" ] && [ -f "$NEW" ] || { echo "missing $OLD or $NEW" >&2; exit 2; }
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$TMP/old" "$TMP/new"
cp "$OLD" "$TMP/old/libh2mi.so"
cp "$NEW" "$TMP/new/libh2mi.so"
make -C examples -s
py() { H2MI_LIBRARY="$TMP/$V/libh2mi.so" python3 "$@" 2>/dev/null; }
cx() { LD_LIBRARY_PATH="$TMP/$V:${LD_LIBRARY_PATH:-}" "$@"; }
step() { py bench.py --no-cpu-baseline --no-create-proof --no-msm-only | tail -1 | python3 -c "
import json,sys
s=sys.stdin.read(); d=json.loads(s[s.index('{'):])
print('step', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['issue_roofline']['frac'])"; }
for r in $(seq "$REP"); do
  for V in old new; do
    echo "== $V"
    case "$MODE" in
      msm) py tools/msm_sweep.py 20 | cut -c1-75; step ;;
      proof) py tools/proof_loop.py 20 12 | tail -1 | cut -c1-90; H2MI_PROOFS=12 cx ./examples/standard_plonk 20 0x5ec2e7 5 1 | grep steady ;;
      small) for k in 5 8 16 20; do H2MI_PROOFS=30 cx ./examples/standard_plonk $k 0x5ec2e7 5 1 | grep steady | awk '{printf "k%s %s  ", "'$k'", $2}'; done; echo ;;
      range) H2MI_PROOFS=5 cx ./examples/halo2_lib range 22 16 77 0x5ec2e7 1 | grep steady; H2MI_PROOFS=8 cx ./examples/halo2_lib poseidon 20 0 5 0x5ec2e7 1 | grep steady ;;
      poly) py tools/poly_sweep.py 20 22 | grep log_n | cut -c1-200 ;;
      evalh) py tools/poly_sweep.py 20 | grep evaluate_h | cut -c1-140 ;;
      *) echo "unknown mode $MODE" >&2; exit 2 ;;
    esac
  done
done
