#!/bin/bash
# the replay bench lines of the other shapes / sizes / distributions (one JSON line each) -> gpurun_out/other_shapes.jsonl
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
OUT=gpurun_out/other_shapes.jsonl
: > $OUT
B="python3 bench.py --no-cpu-baseline --no-create-proof --steps 5 --warmup 2"
$B --dist witness 2>/dev/null | tail -1 >> $OUT
$B --shape halo2_lib_gate 2>/dev/null | tail -1 >> $OUT
$B --shape range_lookup --k 22 2>/dev/null | tail -1 >> $OUT
$B --k 16 2>/dev/null | tail -1 >> $OUT
$B --k 8 2>/dev/null | tail -1 >> $OUT
wc -l $OUT
