#!/usr/bin/env bash
# A/B of ONE environment knob of the -DH2MI_AB build on the C++ host's steady proofs, alternating on one box:
#   tools/ab_env_cpp.sh KNOB=VALUE [ROUNDS] [PROOFS] [sizes...]
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
KNOB=${1:?KNOB=VALUE}; R=${2:-6}; N=${3:-30}; shift 3 || true
SIZES=${*:-"5 8 16 20"}
make -C examples -s
TMP=$(mktemp -d); trap 'rm -rf "$TMP"' EXIT
cp halo2-scaffold_amd/libh2mi_ab.so "$TMP/libh2mi.so"
out=$(mktemp)
for r in $(seq $R); do
  for k in $SIZES; do
    a=$(LD_LIBRARY_PATH="$TMP" H2MI_PROOFS=$N ./examples/standard_plonk $k 0x5ec2e7 5 1 2>/dev/null | grep steady | awk '{print $2}')
    b=$(env "$KNOB" LD_LIBRARY_PATH="$TMP" H2MI_PROOFS=$N ./examples/standard_plonk $k 0x5ec2e7 5 1 2>/dev/null | grep steady | awk '{print $2}')
    echo "k=$k base $a knob $b" | tee -a $out
  done
done
python3 - $out <<'P'
import sys, collections, statistics
d = collections.defaultdict(lambda: ([], []))
for line in open(sys.argv[1]):
    k, _, a, _, b = line.split()
    d[k][0].append(float(a)); d[k][1].append(float(b))
for k, (a, b) in d.items():
    print(f"# {k}: base median {statistics.median(a):.3f}  knob median {statistics.median(b):.3f}")
P
