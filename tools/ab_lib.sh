#!/bin/bash
# A/B of two builds of libh2mi.so on one box, interleaved (box-to-box variation is +-4 %): put the build to compare against next to the
# library as halo2-scaffold_amd/libh2mi.so.prev (git-ignored, travels with gpurun) and run `gpurun -- bash tools/ab_lib.sh`.
cd $GRAFT_REPO_ROOT
L=halo2-scaffold_amd/libh2mi.so
cp $L /tmp/new.so; cp $L.prev /tmp/old.so
for r in 1 2 3; do
  for v in old new; do
    cp /tmp/$v.so $L
    echo "== $v"; python3 tools/msm_sweep.py 20 2>/dev/null | cut -c1-75
    python3 bench.py --no-cpu-baseline --no-create-proof 2>/dev/null | tail -1 | python3 -c "
import json,sys
s=sys.stdin.read(); d=json.loads(s[s.index('{'):])
print('step', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done
cp /tmp/new.so $L
