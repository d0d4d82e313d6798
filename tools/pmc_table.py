"""Developer aid: per-kernel means of every counter found under <prefix>_*/ (tools/pmc_passes.sh)."""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k: continue
    print(k[:100])
    for c in sorted(v): print(f"   {c:34s} {sum(v[c]) / len(v[c]):16.1f}  (n={len(v[c])})")
