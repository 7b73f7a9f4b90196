"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "at::native" in k or "rocclr" in k: continue
        import re
        m = re.search(r"namespace\)::(\w+(<[^>]*>)?)", k)
        k = m.group(1) if m else k[:60]
        acc[k][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
for k, cs in acc.items():
    print(k)
    for c, vals in sorted(cs.items()):
        d = collections.defaultdict(float)
        for did, v in vals: d[did] += v      # sum over XCDs/instances
        vs = list(d.values())
        print("   %-24s mean/dispatch %.4g  (n=%d)" % (c, sum(vs) / len(vs), len(vs)))
