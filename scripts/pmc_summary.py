"""Aggregate a rocprofv3 counter_collection csv per kernel name (runs on the GPU box; prints a small table)."""
import collections, csv, glob, sys
d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:int(sys.argv[3]) if len(sys.argv) > 3 else 14]
for k, v in rows:
    print(k, {c: "%.4g avg over %d" % (v[c] / cnt[k][c], cnt[k][c]) for c in v})
if len(sys.argv) > 2 and sys.argv[2] != "-":      # machine-readable copy: {kernel: {counter: average per launch, "launches": n}}
    import json
    out = {k: dict({c: v[c] / cnt[k][c] for c in v}, launches=max(cnt[k].values())) for k, v in agg.items()}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
