#!/bin/bash
# rocprofv3 kernel-trace timelines of one create_proof: Python host at 2^8 / 2^16 / 2^20, C++ host at 2^8 / 2^20 -> gpurun_out/tl/*.txt
# (tools/trace_timeline.py; copy into profiles/rNN_proof_timeline_*.txt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
make -C examples -s
rm -rf gpurun_out/tl && mkdir -p gpurun_out/tl
for k in 8 16 20; do
  rocprofv3 --kernel-trace -d gpurun_out/tl/py$k -o t --output-format csv -- python3 tools/proof_loop.py $k 6 > gpurun_out/tl/py$k.out 2> gpurun_out/tl/py$k.err
  L=$(python3 -c "import json;print(json.loads(open('gpurun_out/tl/py$k.out').read().strip().splitlines()[-1])['launches_per_proof'])")
  F=$(find gpurun_out/tl/py$k -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_timeline.py $F $L > gpurun_out/tl/proof_timeline_k$k.txt
  tail -1 gpurun_out/tl/py$k.out | cut -c1-100; grep "^# [0-9]* dispatches" gpurun_out/tl/proof_timeline_k$k.txt
  rm -rf gpurun_out/tl/py$k
done
for k in 8 20; do
  H2MI_PROOFS=6 rocprofv3 --kernel-trace -d gpurun_out/tl/cpp$k -o t --output-format csv -- ./examples/standard_plonk $k 0x5ec2e7 5 1 > gpurun_out/tl/cpp$k.out 2> gpurun_out/tl/cpp$k.err
  grep steady gpurun_out/tl/cpp$k.out
  F=$(find gpurun_out/tl/cpp$k -name "*kernel_trace.csv" | head -1)
  L=$(grep -c . gpurun_out/tl/proof_timeline_k$k.txt); L=$(grep "^# [0-9]* dispatches" gpurun_out/tl/proof_timeline_k$k.txt | awk '{print $2}')
  python3 tools/trace_timeline.py $F $L --last > gpurun_out/tl/proof_timeline_cpp_host_k$k.txt
  grep "^# [0-9]* dispatches" gpurun_out/tl/proof_timeline_cpp_host_k$k.txt
  rm -rf gpurun_out/tl/cpp$k
done
