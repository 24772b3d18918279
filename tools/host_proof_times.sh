#!/usr/bin/env bash
# steady-state create_proof times of the C++ host over the prover ABI (ms per proof), the reference's shapes; run on the GPU box:
#   bash tools/host_proof_times.sh > gpurun_out/host_proof_times.txt
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
make -C examples -s
for rep in 1 2 3; do
  for k in 5 8 16 20; do
    H2MI_PROOFS=$([ $k -ge 16 ] && echo 40 || echo 200) ./examples/standard_plonk $k | awk -v k=$k '/steady_ms_per_proof/ {print "standard_plonk k=" k, $2}'
  done
  H2MI_PROOFS=40 ./examples/halo2_lib halo2_lib 20 | awk '/steady_ms_per_proof/ {print "halo2_lib k=20", $2}'
  H2MI_PROOFS=40 ./examples/halo2_lib poseidon 20 | awk '/steady_ms_per_proof/ {print "poseidon k=20", $2}'
  H2MI_PROOFS=200 ./examples/halo2_lib range 13 8 | awk '/steady_ms_per_proof/ {print "range k=13", $2}'
  H2MI_PROOFS=200 ./examples/halo2_lib poseidon 11 | awk '/steady_ms_per_proof/ {print "poseidon k=11 (4 gate columns)", $2}'
done
H2MI_PROOFS=10 ./examples/halo2_lib range 22 16 | awk '/steady_ms_per_proof/ {print "range k=22", $2}'
