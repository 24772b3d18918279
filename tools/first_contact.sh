#!/usr/bin/env bash
# First contact with a multi-GPU node: everything N > 1 that has only ever been rehearsed (gloo, virtual devices, one-rank RCCL) in ONE
# run, logs under gpurun_out/first_contact/ — the two tests a one-GPU box skips (physical peer copies, per-thread device binding on
# device 1), then bench.py at N = 2, 4, 8 in both deployments (one process per GPU over RCCL with the device-resident and the
# host-staged combine; one process driving N devices).  Nothing here is tuned for; it only has to be correct and is reported as measured.
#     bash tools/first_contact.sh [MAX_GPUS]
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/first_contact
mkdir -p "$OUT"
HAVE=$(python3 -c 'import torch; print(torch.cuda.device_count())')
MAX=${1:-$HAVE}
echo "devices visible: $HAVE (running up to N = $MAX)" | tee "$OUT/summary.txt"
if [ "$HAVE" -lt 2 ]; then echo "needs at least two GPUs" | tee -a "$OUT/summary.txt"; exit 2; fi
echo "== the two-GPU tests" | tee -a "$OUT/summary.txt"
timeout -k 10 1500 python3 -m pytest tests/test_gpu_multidevice.py -m gpu -q -rs > "$OUT/pytest_multidevice.log" 2>&1
tail -3 "$OUT/pytest_multidevice.log" | tee -a "$OUT/summary.txt"
PORT=29611
for N in 2 4 8; do
  [ "$N" -gt "$MAX" ] && break
  for MODE in rccl host; do
    echo "== one process per GPU, N = $N, H2MI_COMBINE=$MODE" | tee -a "$OUT/summary.txt"
    PORT=$((PORT + 1))
    H2MI_COMBINE=$MODE timeout -k 10 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
      bench.py --gpus "$N" --steps 10 --warmup 3 > "$OUT/bench_ranks_${N}_${MODE}.json" 2> "$OUT/bench_ranks_${N}_${MODE}.err" \
      || echo "   FAILED (rc $?): see $OUT/bench_ranks_${N}_${MODE}.err" | tee -a "$OUT/summary.txt"
  done
  echo "== one process, $N devices" | tee -a "$OUT/summary.txt"
  timeout -k 10 900 python3 bench.py --gpus "$N" --single-process --steps 10 --warmup 3 > "$OUT/bench_single_${N}.json" 2> "$OUT/bench_single_${N}.err" \
    || echo "   FAILED (rc $?): see $OUT/bench_single_${N}.err" | tee -a "$OUT/summary.txt"
done
echo "== a many-column halo2-lib proof over the sliced SRS (24 range checks at DEGREE 7: 11 + 4 columns, up to 15 partial points per phase), N = 2 over RCCL" | tee -a "$OUT/summary.txt"
python3 tools/flex_proof.py --shape range --k 7 --lookup-bits 4 --count 24 --configure --proofs 2 > "$OUT/flex_wide_1.json" 2> "$OUT/flex_wide_1.err"
PORT=$((PORT + 1))
timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port "$PORT" \
  tools/flex_proof.py --shape range --k 7 --lookup-bits 4 --count 24 --configure --proofs 2 --gpus 2 > "$OUT/flex_wide_2.json" 2> "$OUT/flex_wide_2.err" \
  || echo "   FAILED (rc $?): see $OUT/flex_wide_2.err" | tee -a "$OUT/summary.txt"
python3 - "$OUT" <<'P' | tee -a "$OUT/summary.txt"
import json, os, sys
try:
    a, b = (json.loads([l for l in open(os.path.join(sys.argv[1], f"flex_wide_{i}.json")) if l.startswith("{")][-1]) for i in (1, 2))
    print("wide sliced proof:", "SAME BYTES as the single-GPU proof" if a["proof_sha256"] == b["proof_sha256"] else "DIFFERS from the single-GPU proof",
          f"({b['combines_per_proof']} combines per proof)")
except Exception as e:  # noqa: BLE001
    print("wide sliced proof: no result", e)
P
python3 - "$OUT" <<'P' | tee -a "$OUT/summary.txt"
import glob, json, os, sys
base = None
for path in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(f"{os.path.basename(path):34s} no result line ({e})")
        continue
    cp = d.get("create_proof") or {}
    print(f"{os.path.basename(path):34s} n_gpus {d['n_gpus']}  step {d['ms_per_step']:8.3f} ms  msm_only {d.get('msm_only_ms')} ms  "
          f"speed-up vs 1 {d.get('msm_only_speedup_vs_1')}  create_proof {cp.get('ms_per_proof')} ms  commitments {d['commitments_sha256'][:12]}")
P
