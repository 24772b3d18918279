#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py, then FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes (TCC slots do not fit both), plus the FETCH_SIZE calibration run.
# Outputs land under gpurun_out/prof_$TAG; tools/summarize_profiles.py turns them into profiles/*.
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
[ -x tools/hbm_calib ] || hipcc --offload-arch=gfx950 -O3 tools/hbm_calib.hip -o tools/hbm_calib
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-create-proof --no-msm-only"  # the replay only (its registration aside: summarize_profiles.py separates those two launches)
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- $BENCH > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- $BENCH > $OUT/bench_write.json 2> $OUT/write.err
echo "write done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/calib -o c --output-format csv -- ./tools/hbm_calib > $OUT/calib.txt 2> $OUT/calib.err
echo "calib done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/sq -o s --output-format csv -- $BENCH > $OUT/bench_sq.json 2> $OUT/sq.err
echo "sq done"
# drop the per-dispatch traces of the counter passes except what the summary needs (size)
ls -la $OUT/*
