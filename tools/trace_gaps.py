"""Idle time of the GPU between kernels, from a rocprofv3 --kernel-trace CSV of a bench run:
python tools/trace_gaps.py <kernel_trace.csv> [fraction of the run to skip at the start, default 0.5] [last ms to analyse].
Prints, for the steady-state tail of the run: span, union of kernel intervals (all queues), idle = span - union, the gap
histogram and the kernels after which the largest share of the idle time sits."""
import collections, csv, sys

path = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t0 + int((t1 - t0) * skip)
if len(sys.argv) > 3:                                     # third argument: analyse the last <ms> of the run instead
    cut = t1 - int(float(sys.argv[3]) * 1e6)
rows = [r for r in rows if r[0] >= cut]
span = max(r[1] for r in rows) - rows[0][0]
busy = 0
gaps = []
end = rows[0][0]
last = None
overlap = 0
for s, e, name, q in rows:
    if s > end:
        gaps.append((s - end, last, name))
        busy += e - s
        end = e
        last = name
    else:
        if e > end:
            busy += e - end
            end = e
            last = name
        overlap += 1
idle = span - busy
print("kernels %d (overlapping a previous one: %d), span %.2f ms, busy %.2f ms, idle %.2f ms = %.1f %%"
      % (len(rows), overlap, span / 1e6, busy / 1e6, idle / 1e6, 100.0 * idle / span))
h = collections.Counter()
for g, _, _ in gaps:
    b = 1
    while b < g / 1000.0:
        b *= 2
    h[b] += g
print("idle time by gap length (us bucket upper bound: ms):", {k: round(v / 1e6, 2) for k, v in sorted(h.items())})
after = collections.Counter()
cnt = collections.Counter()
for g, prev, nxt in gaps:
    key = (prev or "")[:60]
    after[key] += g
    cnt[key] += 1
print("idle time after kernel (ms, gaps, mean us):")
for k, v in after.most_common(25):
    print("  %-62s %7.2f %5d %6.1f" % (k, v / 1e6, cnt[k], v / cnt[k] / 1e3))
before = collections.Counter()
for g, prev, nxt in gaps:
    before[(nxt or "")[:60]] += g
print("idle time before kernel (ms):")
for k, v in before.most_common(15):
    print("  %-62s %7.2f" % (k, v / 1e6))
