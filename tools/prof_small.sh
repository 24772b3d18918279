#!/usr/bin/env bash
# SQ counters of the small-set MSM kernels (where does a level's 3.2 us go: VALU issue, LDS, barriers?): gpurun_out/prof_small/summary.txt
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
OUT=gpurun_out/prof_small; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/sq -o s --output-format csv -- python3 tools/msm_b2b.py 8 22 > /dev/null 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA -d $OUT/sq2 -o s --output-format csv -- python3 tools/msm_b2b.py 8 22 > /dev/null 2> $OUT/sq2.err
python3 - <<PY
import csv, glob, collections
res = collections.defaultdict(dict)
for sub in ("sq", "sq2"):
    for f in glob.glob(f"$OUT/{sub}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "small" not in k: continue
            acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for (k, c), v in acc.items(): res[k][c] = v / cnt[(k, c)]
with open("$OUT/summary.txt", "w") as fo:
    for k, d in res.items():
        fo.write(k + "\n")
        for c, v in sorted(d.items()): fo.write(f"   {c:26s} {v:14.1f}\n")
print(open("$OUT/summary.txt").read())
PY
