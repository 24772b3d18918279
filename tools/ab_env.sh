#!/bin/bash
# A/B of one environment knob on the replay bench, interleaved on one box: tools/ab_env.sh VAR "v1 v2" [repeats] [bench args]
VAR=$1; VALS=$2; REP=${3:-2}; shift 3
for r in $(seq $REP); do
  for v in $VALS; do
    env $VAR=$v python3 bench.py --no-cpu-baseline --no-create-proof "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
s=sys.stdin.read(); d=json.loads(s[s.index('{'):])
print('$VAR=$v', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['issue_roofline']['frac'])"
  done
done
