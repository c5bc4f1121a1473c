"""Sums rocprofv3 --pmc counter values per kernel name.  python tools/pmc_by_kernel.py <dir with *counter_collection.csv> [name filter]"""
import sys, csv, glob, os, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), k)
        if key not in seen:
            seen.add(key); cnt[k] += 1
for k, d in sorted(tot.items()):
    print(k, "dispatches", cnt[k], {c: f"{v / max(cnt[k], 1):.4g}" for c, v in sorted(d.items())})
