cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for c in 16 17; do
    echo "== c=$c"
    H2MI_MSM_C=$c python3 tools/proof_loop.py 20 12 2>/dev/null | tail -1 | cut -c1-160
    H2MI_MSM_C=$c python3 bench.py --no-cpu-baseline --no-create-proof 2>/dev/null | tail -1 | python3 -c "
import json,sys
s=sys.stdin.read(); d=json.loads(s[s.index('{'):])
print('step', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done
