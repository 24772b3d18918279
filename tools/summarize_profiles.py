#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/prof_TAG) into the committed
summaries under profiles/: kernel stats CSV, per-kernel HBM traffic (FETCH_SIZE corrected by the
calibration run as MI355X_MICROARCH.md prescribes, WRITE_SIZE as read), SQ counter summary."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
shutil.copy(f"{src}/trace/t_kernel_stats.csv", f"profiles/{tag}_kernel_stats.csv")


def per_kernel(path, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("h2::", "")
        if "rocprim" in k:
            k = "rocprim_scan"
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


# calibration: FETCH_SIZE is reported in KiB-ish units of 1024 B per count on this stack; derive the
# factor from the known-byte kernels instead of trusting a unit.
cal = per_kernel(f"{src}/calib/c_counter_collection.csv", "FETCH_SIZE")
known_gather = (1 << 18) * 64 * 64 + (1 << 18) * 64 * 4
known_stream = (1 << 24) * 64
f_gather = known_gather / cal["calib_gather64"][0]
f_stream = known_stream / cal["calib_stream16"][0]
fetch = per_kernel(f"{src}/fetch/f_counter_collection.csv", "FETCH_SIZE")
write = per_kernel(f"{src}/write/w_counter_collection.csv", "WRITE_SIZE")
out = {
    "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 3 --warmup 1`, {tag}",
    "fetch_bytes_per_count": {"gather64_pattern": f_gather, "stream16_pattern": f_stream,
                              "note": "bytes per FETCH_SIZE count from tools/hbm_calib (known byte counts); "
                                      "stream16 is the guide's wide-coalesced case (expect 2048 = 2 x 1024)"},
    "kernels": {},
}
for k in sorted(fetch):
    fcount, n = fetch[k]
    wcount = write.get(k, (0, 0))[0]
    factor = f_gather if k in ("k_msm_accum",) else f_stream
    out["kernels"][k] = {"launches": n, "FETCH_SIZE_avg": fcount, "WRITE_SIZE_avg": wcount,
                         "fetch_bytes_per_launch": fcount * factor, "write_bytes_per_launch": wcount * 1024,
                         "fetch_factor_used": "gather64" if factor == f_gather else "stream16"}
acc = out["kernels"].get("k_msm_accum")
if acc:
    out["k_msm_accum_hbm_bytes_per_launch"] = int(acc["fetch_bytes_per_launch"] + acc["write_bytes_per_launch"])
json.dump(out, open(f"profiles/{tag}_traffic.json", "w"), indent=1)

sq = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f"{src}/sq/s_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0].replace("h2::", "")
    if "rocprim" in k:
        k = "rocprim_scan"
    sq[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        n[k] += 1
with open(f"profiles/{tag}_sq_counters.csv", "w") as f:
    names = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_LDS", "GRBM_GUI_ACTIVE"]
    f.write("kernel,launches," + ",".join(x + "_per_launch" for x in names) + "\n")
    for k in sorted(sq):
        f.write(k + f",{n[k]}," + ",".join(f"{sq[k][x] / max(n[k], 1):.0f}" for x in names) + "\n")
# k_msm_accum launch by launch (round-4 VERDICT: the stats CSV's average mixes the replay's dense 2^20 MSMs with the registration's
# all-ones MSMs): the kernel trace's own durations, classed by size.  Every launch has the same grid (a resident grid of 131072 threads), so
# the class is read off the duration: the all-ones MSMs of the two registrations finish in < 0.1 ms (the dominant-value shift empties them),
# every other launch is a dense 2^20 MSM of the replay — including the ones that share the SIMDs with transform passes and take 1.7 - 2.6 ms
import glob
import statistics

tr = glob.glob(f"{src}/trace/*kernel_trace.csv")
if tr:
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(tr[0])) if "k_msm_accum" in r["Kernel_Name"]]
    if durs:
        med = statistics.median(durs)
        replay = [d for d in durs if d >= 0.5 * med]
        other = [d for d in durs if d < 0.5 * med]
        alone = [d for d in replay if d <= 1.3 * med]
        bench_line = json.loads(open(f"{src}/bench_trace.json").read().strip().splitlines()[-1])
        acc = {
            "source": f"rocprofv3 --kernel-trace over `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-create-proof --no-msm-only`, {tag}: every k_msm_accum launch",
            "replay_dense_launches": {"count": len(replay), "avg_ns": round(sum(replay) / len(replay)), "min_ns": min(replay), "max_ns": max(replay)},
            "of_which_not_sharing_the_chip_with_transforms": {"count": len(alone), "avg_ns": round(sum(alone) / len(alone)),
                                                              "what": "launches within 30 % of the median: the accumulation alone on the SIMDs"},
            "other_launches": {"count": len(other), "what": "the registration's all-ones MSMs (one per base set: the sum point of the dominant-value shift)",
                               "durations_ns": sorted(other)},
            "algorithmic_bytes_per_launch": 96 << 20,
            "hbm_frac_from_trace": round((96 << 20) / (sum(replay) / len(replay) * 1e-9) / 8e12, 6),
            "bench_line_in_this_run": {"avg_launch_ms": bench_line["roofline"]["avg_launch_ms"], "frac": bench_line["roofline"]["frac"]},
        }
        json.dump(acc, open(f"profiles/{tag}_accum_launches.json", "w"), indent=1)
        print("k_msm_accum replay launches", acc["replay_dense_launches"], "frac", acc["hbm_frac_from_trace"], "bench", acc["bench_line_in_this_run"])
for fn in ("bench_trace.json",):
    shutil.copy(f"{src}/{fn}", f"profiles/{tag}_{fn}")
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1))
print({k: (round(v["fetch_bytes_per_launch"] / 1e6, 1), round(v["write_bytes_per_launch"] / 1e6, 1)) for k, v in out["kernels"].items()})
