#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> timeline of the last-but-one group of LAUNCHES dispatches (the last timed proof of
tools/proof_loop.py): start (us from the group's first dispatch), duration, gap to the latest earlier end on the same queue,
queue id, kernel.  Ends with the totals that matter for a latency-bound proof: span, sum of durations per queue, idle time
of the union of all queues.
Usage: trace_timeline.py TRACE.csv LAUNCHES [--summary] [--last]   (--last: the final LAUNCHES dispatches, e.g. the C++ examples' steady loop)"""
import csv
import re
import sys

path, launches = sys.argv[1], int(sys.argv[2])
rows = [r for r in csv.DictReader(open(path)) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
group = rows[-launches:] if "--last" in sys.argv or len(rows) < 2 * launches else rows[-2 * launches : -launches]
t0 = int(group[0]["Start_Timestamp"])
last_end = {}
intervals = []
per_kernel = {}
for r in group:
    s, e, q = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Queue_Id"]
    name = re.sub(r"^h2::", "", r["Kernel_Name"].split("(")[0]).replace("void ", "")
    gap = s - last_end.get(q, s)
    last_end[q] = e
    intervals.append((s, e))
    d = per_kernel.setdefault(name, [0, 0.0])
    d[0] += 1
    d[1] += (e - s) / 1e3
    if "--summary" not in sys.argv:
        print(f"{s / 1e3:10.1f} {(e - s) / 1e3:9.1f} {gap / 1e3:8.1f}  q{q:>2}  {name}")
span = max(e for _, e in intervals)
intervals.sort()
busy, cur_s, cur_e = 0, *intervals[0]
for s, e in intervals[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"# {len(group)} dispatches, span {span / 1e3:.1f} us, some kernel running {busy / 1e3:.1f} us, device idle {(span - busy) / 1e3:.1f} us")
for name, (c, us) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"# {us:9.1f} us {c:4d} x {name}")
