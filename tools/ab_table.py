import sys, json, collections
rows = collections.OrderedDict(); cur = None; cols = []
for line in sys.stdin:
    line = line.strip()
    if line.startswith("=="):
        cur = line[3:]
        if cur not in cols: cols.append(cur)
        continue
    if not line.startswith("{"): continue
    n = int(line.split('"n": ')[1].split(",")[0]); m = line.split('"map": "')[1].split('"')[0]; ms = float(line.split('"ms": ')[1].split(",")[0])
    rows.setdefault((n, m), collections.defaultdict(list))[cur].append(ms)
print("%-22s" % "case" + "".join("%-18s" % c for c in cols))
for (n, m), d in rows.items():
    print("%-22s" % ("%dq %s" % (n, m)) + "".join("%-18s" % "/".join("%.3f" % x for x in d[c]) for c in cols))
