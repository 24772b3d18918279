#!/usr/bin/env bash
# HBM traffic of the NTT passes, stand-alone (tools/ntt_sweep.py at 2^20): FETCH_SIZE / WRITE_SIZE in separate --pmc passes, for
# the product (tile-ordered inter-pass twiddle matrices) and, through the -DH2MI_AB library, for the round-3 form
# (H2MI_NTT_NO_WMAT=1: gather from the full power table).  Output: gpurun_out/prof_ntt_traffic/summary.txt
set -euo pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
OUT=gpurun_out/prof_ntt_traffic
mkdir -p $OUT
run() {  # $1 = tag
  rocprofv3 --kernel-trace --stats -d $OUT/$1/trace -o t --output-format csv -- python3 tools/ntt_sweep.py 20 21 > $OUT/$1.sweep.txt 2> $OUT/$1.trace.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/$1/fetch -o f --output-format csv -- python3 tools/ntt_sweep.py 20 21 > /dev/null 2> $OUT/$1.fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/$1/write -o w --output-format csv -- python3 tools/ntt_sweep.py 20 21 > /dev/null 2> $OUT/$1.write.err
}
run wmat
export H2MI_LIBRARY="$PWD/halo2-scaffold_amd/libh2mi_ab.so" H2MI_NTT_NO_WMAT=1
run gather
unset H2MI_LIBRARY H2MI_NTT_NO_WMAT
[ -x tools/hbm_calib ] || hipcc --offload-arch=gfx950 -O3 tools/hbm_calib.hip -o tools/hbm_calib
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/calib -o c --output-format csv -- ./tools/hbm_calib > $OUT/calib.txt 2> $OUT/calib.err
python3 - <<PY
import csv, glob, collections
out = "$OUT"
def agg(path):
    acc = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    return {k: (v / cnt[k], cnt[k]) for k, v in acc.items()}
with open(f"{out}/summary.txt", "w") as fo:
    for tag in ("wmat", "gather", "calib"):
        fo.write(f"== {tag}\n")
        for sub in ("fetch", "write", ""):
            for f in glob.glob(f"{out}/{tag}/{sub}/**/*counter_collection.csv", recursive=True) if tag != "calib" or sub == "" else []:
                for (k, c), (v, n) in sorted(agg(f).items()):
                    if "ntt" in k or "calib" in k or "gather" in k or "stream" in k:
                        fo.write(f"  {k:48s} {c:12s} avg {v:14.1f} per launch over {n} launches\n")
        for f in glob.glob(f"{out}/{tag}/trace/**/*kernel_stats.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "ntt" in r["Name"]:
                    fo.write(f"  {r['Name'].split('(')[0]:48s} calls {r['Calls']:>5s} avg_ns {r['AverageNs']}\n")
        if tag != "calib":
            fo.write(open(f"{out}/{tag}.sweep.txt").read())
    fo.write(open(f"{out}/calib.txt").read())
print(open(f"{out}/summary.txt").read())
PY
