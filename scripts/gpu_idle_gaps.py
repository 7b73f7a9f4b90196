"""GPU idle time inside a bench step, from a rocprofv3 --kernel-trace CSV: the union of kernel intervals over the last
full step (between two hook_pool bursts), the idle gaps above a threshold and the kernels on either side of the largest.
argv: kernel_trace.csv"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# step boundaries: the CSV-writing gap is the longest idle stretch; take the stretch between the last two such gaps
gaps = [(rows[i + 1][0] - max(r[1] for r in rows[max(0, i - 50):i + 1]), i) for i in range(len(rows) - 1)]
big = sorted(g for g in gaps if g[0] > 3_000_000)          # > 3 ms idle: end of a step (CSV + cache files)
cut = sorted(i for _, i in big)[-2:] if len(big) >= 2 else [0, len(rows) - 2]
lo, hi = cut[0] + 1, cut[1] + 1
step = rows[lo:hi]
t0, t1 = step[0][0], max(r[1] for r in step)
busy, end = 0, t0
idle = []
for s, e, n in step:
    if s > end:
        idle.append((s - end, n))
        busy += e - s
        end = e
    elif e > end:
        busy += e - end
        end = e
print("step: %d kernels, span %.1f ms, GPU busy %.1f ms (%.1f %%)" % (len(step), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0)))
idle.sort(reverse=True)
tot = sum(g for g, _ in idle)
print("idle inside the span: %.2f ms in %d gaps; gaps > 20 us: %.2f ms" % (tot / 1e6, len(idle), sum(g for g, _ in idle if g > 20000) / 1e6))
for g, n in idle[:12]:
    print("  %8.1f us before %s" % (g / 1e3, n[:90]))
