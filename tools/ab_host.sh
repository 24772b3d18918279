#!/usr/bin/env bash
# A/B of the C++ host on one GPU box: examples/ab_old/* (built on the dev box from an earlier commit: its include/ and, when examples/ab_old/lib/libh2mi.so
# exists, its library through the binaries' RUNPATH; see
# DESIGN 4.5) against the current examples, alternating, steady-state ms per proof.  tools/ab_host.sh [ROUNDS] [PROOFS]
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
R=${1:-3}; N=${2:-40}
make -C examples -s
mkdir -p gpurun_out/ab_host
out=gpurun_out/ab_host/result.txt; : > $out
run() { # label binary args...
  local label=$1; shift
  local ms; ms=$(H2MI_PROOFS=$N "$@" 2>/dev/null | grep steady | awk '{print $2}')
  echo "$label $ms" | tee -a $out
}
KS=${3:-"5 8 16 20"}
for r in $(seq $R); do
  for k in $KS; do
    run "old standard_plonk k=$k" ./examples/ab_old/standard_plonk $k 0x5ec2e7 5 1
    run "new standard_plonk k=$k" ./examples/standard_plonk $k 0x5ec2e7 5 1
  done
done
# the halo2-lib builders' prover (include/h2mi_flex.hpp): FLEX="shape:k:bits ..."
for r in $(seq $R); do
  for f in ${FLEX:-}; do
    IFS=: read shape k bits <<< "$f"
    run "old $shape k=$k" ./examples/ab_old/halo2_lib $shape $k $bits
    run "new $shape k=$k" ./examples/halo2_lib $shape $k $bits
  done
done
python3 - $out <<'P'
import sys, collections, statistics
d = collections.defaultdict(list)
for line in open(sys.argv[1]):
    *label, ms = line.split()
    d[" ".join(label)].append(float(ms))
for k, v in sorted(d.items(), key=lambda kv: (kv[0].split("k=")[1], kv[0])):
    print(f"# {k}: min {min(v):.3f} median {statistics.median(v):.3f} max {max(v):.3f} ({len(v)} runs)")
P
