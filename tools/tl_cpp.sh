#!/usr/bin/env bash
# rocprofv3 kernel trace of the C++ host's steady proofs at 2^K -> gpurun_out/tl/cpp_k$K.txt (tools/trace_timeline.py): tools/tl_cpp.sh K [LAUNCHES]
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
K=${1:-20}; L=${2:-210}
make -C examples -s
mkdir -p gpurun_out/tl
H2MI_PROOFS=12 rocprofv3 --kernel-trace -d gpurun_out/tl/cpp$K -o t --output-format csv -- ./examples/standard_plonk $K 0x5ec2e7 5 1 > gpurun_out/tl/cpp$K.out 2> gpurun_out/tl/cpp$K.err
grep steady gpurun_out/tl/cpp$K.out
F=$(find gpurun_out/tl/cpp$K -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $F $L --last > gpurun_out/tl/cpp_k$K.txt
grep "^#" gpurun_out/tl/cpp_k$K.txt | head -30
rm -rf gpurun_out/tl/cpp$K
