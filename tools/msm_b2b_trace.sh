#!/usr/bin/env bash
# kernel timeline of 22 back-to-back 2^17 MSMs (what limits the 8-GPU slice rate): gpurun_out/b2b/timeline_k17.txt
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
K=${1:-17}
rm -rf gpurun_out/b2b && mkdir -p gpurun_out/b2b
python3 tools/msm_b2b.py $K 22 | tail -1
rocprofv3 --kernel-trace -d gpurun_out/b2b/tr -o t --output-format csv -- python3 tools/msm_b2b.py $K 22 > gpurun_out/b2b/run.out 2> gpurun_out/b2b/run.err
tail -1 gpurun_out/b2b/run.out
F=$(find gpurun_out/b2b/tr -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $F 420 --last > gpurun_out/b2b/timeline_k$K.txt
tail -22 gpurun_out/b2b/timeline_k$K.txt
rm -rf gpurun_out/b2b/tr
