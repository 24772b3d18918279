#!/bin/bash
# rocprofv3 kernel trace of the C++ host's steady proofs at 2^20 -> gpurun_out/tl/k20_cpp_timeline.txt (tools/trace_timeline.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
make -C examples -s
rm -rf gpurun_out/tl && mkdir -p gpurun_out/tl
H2MI_PROOFS=6 rocprofv3 --kernel-trace -d gpurun_out/tl/cpp -o t --output-format csv -- ./examples/standard_plonk 20 0x5ec2e7 5 1 > gpurun_out/tl/cpp.out 2> gpurun_out/tl/cpp.err
grep steady gpurun_out/tl/cpp.out
F=$(find gpurun_out/tl/cpp -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $F 210 --last > gpurun_out/tl/k20_cpp_timeline.txt
tail -30 gpurun_out/tl/k20_cpp_timeline.txt
rm -rf gpurun_out/tl/cpp
