#!/usr/bin/env bash
# steady-state create_proof times of the C++ host for the many-column halo2-lib shapes (ms per proof); run on the GPU box:
#   bash tools/wide_proof_times.sh > gpurun_out/wide_proof_times.txt
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
make -C examples -s
run() { H2MI_PROOFS=$1 ./examples/halo2_lib "${@:2}" | awk -v what="${*:2}" '/^columns/ {c=$0} /steady_ms_per_proof/ {print what " | " c " | " $2 " ms"}'; }
for rep in 1 2; do
  run 100 poseidon 8
  run 100 poseidon 9
  run 100 poseidon 10
  run 100 poseidon 11
  run 100 range 6 4 12 5ec2e7 11 8
  run 100 range 7 4 12 5ec2e7 11 24
  run 100 range 6 4 12 5ec2e7 11 10 11 8
done
