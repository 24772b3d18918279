import statistics, re, sys
d = {"old": {}, "new": {}}
cur = None
for line in open(sys.argv[1]):
    if line.startswith("=="):
        cur = line.split()[1]
        continue
    for k, v in re.findall(r"k(\d+) ([\d.]+)", line):
        d[cur].setdefault(k, []).append(float(v))
for k in d["old"]:
    print("k", k, "old", statistics.median(d["old"][k]), "new", statistics.median(d["new"][k]))
