"""rocprofv3 --pmc pass outputs (rocpd sqlite under <dir>) -> one JSON of mean counter values per dispatch of the kernel matching <pattern>.

    python3 scripts/pmc_to_json.py <dir> <kernel pattern> <out.json> [key=value ...]     (extra keys are stored verbatim; numbers as numbers)
"""
import collections, glob, json, sqlite3, sys

root, pat, dst = sys.argv[1], sys.argv[2], sys.argv[3]
res = {"_how": "rocprofv3 --kernel-trace --pmc <counters> (one pass per counter group); mean per dispatch, summed over XCD instances", "kernel_pattern": pat}
durs = []
for path in sorted(glob.glob(root + "/**/*.db", recursive=True)):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    ki, ci, vi, di = (cols.index(c) for c in ("kernel_name", "counter_name", "value", "dispatch_id"))
    st, en = cols.index("start"), cols.index("end")
    acc, dur = collections.defaultdict(lambda: collections.defaultdict(float)), {}
    name = None
    for r in db.execute("select * from counters_collection"):
        if pat not in r[ki]:
            continue
        name = r[ki]
        acc[r[ci]][r[di]] += r[vi]
        dur[r[di]] = r[en] - r[st]
    for c, d in acc.items():
        res[c] = sum(d.values()) / len(d)
    if dur:
        durs.append(sum(dur.values()) / len(dur) / 1e3)
        res["kernel"] = name.split("(anonymous namespace)::")[-1][:120]
if durs:
    res["mean_dur_us_under_counters"] = round(sum(durs) / len(durs), 1)
for kv in sys.argv[4:]:
    k, v = kv.split("=", 1)
    try:
        res[k] = float(v) if "." in v or "e" in v.lower() else int(v)
    except ValueError:
        res[k] = v
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res, indent=1))
