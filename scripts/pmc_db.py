"""Summarise rocprofv3 --pmc results (rocpd sqlite): per kernel matching argv[2], mean counter value per dispatch."""
import sqlite3, sys, glob, collections
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for path in sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    rows = db.execute("select * from counters_collection").fetchall()
    ki, ci, vi, di = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value"), cols.index("dispatch_id")
    st, en = cols.index("start"), cols.index("end")
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = {}
    for r in rows:
        if pat not in r[ki]: continue
        acc[r[ci]][r[di]] += r[vi]
        dur[r[di]] = r[en] - r[st]
    print(path, " dispatches:", len(dur), " mean dur us: %.1f" % (sum(dur.values()) / max(1, len(dur)) / 1e3))
    for c, d in sorted(acc.items()):
        print("   %-24s %.5g" % (c, sum(d.values()) / len(d)))
