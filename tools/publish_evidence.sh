#!/usr/bin/env bash
# copies what tools/final_evidence.sh TAG left under gpurun_out/ into profiles/TAG_* (run in the build container, repo root)
set -euo pipefail
TAG=${1:-r05}
E=gpurun_out/ev_$TAG
python3 tools/summarize_profiles.py $TAG > /dev/null
cp $E/bench.json profiles/${TAG}_bench.json
cp $E/kernel_stats_with_provers.csv profiles/${TAG}_kernel_stats_with_provers.csv
cp $E/op_bench.json profiles/${TAG}_op_bench.json
cp $E/poly_sweep.txt profiles/${TAG}_poly_sweep.txt
cp $E/other_shapes.jsonl profiles/${TAG}_bench_other_shapes.json
python3 - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
s = open(f"gpurun_out/ev_{tag}/scaling_model_k20.json").read()
open(f"profiles/{tag}_scaling_model_k20.json", "w").write(json.dumps(json.loads(s[s.index("{"):]), indent=1) + "\n")
PY
for f in proof_timeline_k8 proof_timeline_k16 proof_timeline_k20 proof_timeline_cpp_host_k8 proof_timeline_cpp_host_k20; do cp $E/$f.txt profiles/${TAG}_$f.txt; done
cp $E/poseidon_host_profile.txt profiles/${TAG}_poseidon_host_profile.txt
[ -f $E/g1fft_sweep.txt ] && cp $E/g1fft_sweep.txt profiles/${TAG}_g1fft_sweep.txt
[ -f $E/fuzz.txt ] && cp $E/fuzz.txt profiles/${TAG}_fuzz.txt
cp $E/ntt_sweep.txt profiles/${TAG}_ntt_sweep.txt
{ echo "# MSM sweep, final build (tools/msm_sweep.py: general pipeline 2^18 .. 2^22; tools/msm_small_sweep.py 2^5 .. 2^17: 'default' = the path the library picks"; echo "# by itself (latency path up to 2^14 points; above 2^13 the general pipeline once four MSMs are queued without a join), 'general' = H2MI_MSM_GENERAL,"; echo "# 'small path' = the latency path pinned whatever the queue depth (libh2mi_ab.so, H2MI_MSM_NO_AUTO_STREAM=1))"; cat $E/msm_sweep.txt; echo; grep -v amdgpu $E/msm_small_final.txt; } > profiles/${TAG}_msm_sweep.txt
python3 - "$TAG" <<'PY'
import csv, json, sys
tag = sys.argv[1]
d = json.loads(open(f"profiles/{tag}_bench.json").read().strip().splitlines()[-1])
print("step", d["ms_per_step"], "adds/s %.3g" % d["value"], "frac", d["roofline"]["frac"], "accum_ms", d["roofline"]["avg_launch_ms"], "issue", d["issue_roofline"]["frac"],
      "msm_only", d["msm_only_ms"], "pipeline", d["pipeline_ms_per_step"])
cp = d["create_proof"]
print("k20", cp["ms_per_proof"], cp.get("cpp_host_ms"), *[(k, d[k].get("python_host_ms"), d[k].get("cpp_host_ms")) for k in ("create_proof_k16", "create_proof_k8", "create_proof_k5")])
print({k: (v.get("ms_per_proof"), v.get("cpp_host_ms")) for k, v in d["halo2_lib_create_proof"].items()})
print({k: (v["us_per_transform"], v["hbm"]["frac"], v["mad_issue"]["frac"]) for k, v in d["roofline_ntt"].items() if k != "bound"})
print("cpu", d["cpu_baseline"]["msm_seconds"], d["cpu_baseline"]["projected_step_seconds"])
k = json.load(open(f"profiles/{tag}_traffic.json"))["kernels"]
for name in ("k_msm_accum", "void k_ntt_pass_col<1024u, 10u>", "void k_ntt_pass_row<1024u, 10u>", "void k_ntt_pass_col<1024u, 8u>"):
    if name in k: print(name, round(k[name]["fetch_bytes_per_launch"] / 1e6, 1), round(k[name]["write_bytes_per_launch"] / 1e6, 1), k[name]["launches"])
for r in csv.DictReader(open(f"profiles/{tag}_kernel_stats.csv")):
    if "k_msm_accum" in r["Name"]: print("stats", r["Calls"], r["AverageNs"])
b = json.loads(open(f"profiles/{tag}_bench_trace.json").read().strip().splitlines()[-1])
print("trace", b["roofline"]["avg_launch_ms"], b["ms_per_step"])
for l in open(f"profiles/{tag}_bench_other_shapes.json"):
    x = json.loads(l); print(x["config"]["workload"][:44], x["config"].get("scalar_distribution"), x["ms_per_step"])
PY
