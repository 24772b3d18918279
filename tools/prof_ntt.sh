#!/bin/bash
# rocprofv3 counters for the NTT kernels alone (tools/ntt_sweep.py at one size): SQ issue / wait split, LDS, HBM bytes.
# Runs on the GPU box; outputs under gpurun_out/prof_ntt_$TAG.
set -e
TAG=${1:-r02}
LOGN=${2:-20}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_ntt_$TAG
mkdir -p $OUT
CMD="python3 tools/ntt_sweep.py $LOGN"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- $CMD > $OUT/sweep.txt 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/sq -o s --output-format csv -- $CMD > /dev/null 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVES -d $OUT/sq2 -o s --output-format csv -- $CMD > /dev/null 2> $OUT/sq2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- $CMD > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- $CMD > /dev/null 2> $OUT/write.err
python3 - <<PY
import csv, glob, collections
out = "$OUT"
def agg(path):
    rows = list(csv.DictReader(open(path)))
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    return {k: {c: v / cnt[(k, c)] for c, v in d.items()} for k, d in acc.items()}
res = {}
for sub in ("sq", "sq2", "fetch", "write"):
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for k, d in agg(f).items():
            res.setdefault(k, {}).update(d)
with open(f"{out}/summary.txt", "w") as fo:
    for k, d in res.items():
        if "ntt" not in k: continue
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            fo.write(f"   {c:28s} {v:16.1f}\n")
for f in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
    open(f"{out}/kernel_stats.csv", "w").write(open(f).read())
print(open(f"{out}/summary.txt").read())
PY
cat $OUT/sweep.txt
