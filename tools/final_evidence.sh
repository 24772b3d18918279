#!/usr/bin/env bash
# everything profiles/rNN_* is made from, from ONE build: bash tools/final_evidence.sh r05 [part]   (outputs: gpurun_out/ev_r05/)
# part 1: bench, rocprof passes, op_bench; part 2: sweeps, fuzz, timelines, scaling model (two gpurun calls of < 20 min each); no part: both
set -uo pipefail
TAG=${1:-r05}
PART=${2:-all}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
OUT=gpurun_out/ev_$TAG
mkdir -p $OUT
make -C examples -s
if [ "$PART" = all ] || [ "$PART" = 1 ]; then
echo "== bench"; python3 bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.err
echo "== profiles"; bash tools/collect_profiles.sh $TAG > $OUT/collect.log 2>&1; tail -3 $OUT/collect.log
echo "== with provers"; rocprofv3 --kernel-trace --stats -d $OUT/provers -o t --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_provers.json 2> $OUT/provers.err
cp $(find $OUT/provers -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_with_provers.csv 2>/dev/null; rm -rf $OUT/provers
echo "== op_bench"; python3 tools/op_bench.py > $OUT/op_bench.json 2> $OUT/op_bench.err
fi
if [ "$PART" = all ] || [ "$PART" = 2 ]; then
echo "== small msm"; python3 tools/msm_small_sweep.py 5 8 10 12 13 14 15 16 17 > $OUT/msm_small_final.txt 2>/dev/null
H2MI_LIBRARY=$PWD/halo2-scaffold_amd/libh2mi_ab.so H2MI_MSM_NO_AUTO_STREAM=1 python3 tools/msm_small_sweep.py 12 13 14 15 2>/dev/null | grep "small path" >> $OUT/msm_small_final.txt
echo "== group fft"; python3 tools/g1fft_sweep.py 12 16 18 20 22 2>/dev/null | grep -v amdgpu > $OUT/g1fft_sweep.txt
echo "== fuzz"; python3 tools/fuzz_parity.py 150 2>&1 | grep -v amdgpu > $OUT/fuzz.txt; tail -1 $OUT/fuzz.txt
echo "== msm sweep"; python3 tools/msm_sweep.py 18 19 20 21 22 > $OUT/msm_sweep.txt 2>/dev/null
echo "== ntt sweep"; python3 tools/ntt_sweep.py 16 18 19 20 21 22 24 > $OUT/ntt_sweep.txt 2>/dev/null
echo "== poly sweep"; python3 tools/poly_sweep.py 20 22 > $OUT/poly_sweep.txt 2>/dev/null
echo "== host profile"; PROOF_LOOP_CPROFILE=1 python3 tools/proof_loop.py 20 8 poseidon > $OUT/poseidon_host_profile.txt 2>&1
echo "== other shapes"; bash tools/bench_other_shapes.sh > /dev/null 2>&1; cp gpurun_out/other_shapes.jsonl $OUT/ 2>/dev/null
echo "== scaling model"; python3 tools/scaling_model.py 20 > $OUT/scaling_model_k20.json 2> $OUT/scaling.err
echo "== timelines"; bash tools/tl_all.sh > $OUT/tl.log 2>&1; cp gpurun_out/tl/*.txt $OUT/ 2>/dev/null
fi
echo "== done"; ls $OUT | head -50
