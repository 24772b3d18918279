#!/bin/bash
# rocprofv3 kernel trace of the C++ host's steady range proofs at DEGREE 22 / LOOKUP_BITS 16 -> gpurun_out/tl/k22_range_cpp_timeline.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
make -C examples -s
rm -rf gpurun_out/tl/r && mkdir -p gpurun_out/tl
H2MI_PROOFS=3 rocprofv3 --kernel-trace -d gpurun_out/tl/r -o t --output-format csv -- ./examples/halo2_lib range 22 16 77 0x5ec2e7 1 > gpurun_out/tl/r.out 2> gpurun_out/tl/r.err
grep steady gpurun_out/tl/r.out
F=$(find gpurun_out/tl/r -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $F ${LAUNCHES:-307} --last > gpurun_out/tl/k22_range_cpp_timeline.txt
grep "^#" gpurun_out/tl/k22_range_cpp_timeline.txt
rm -rf gpurun_out/tl/r
