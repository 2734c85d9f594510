"""Idle time between consecutive kernels of a rocprofv3 kernel trace: python scripts/probes/trace_gaps.py <kernel_trace.csv> [min_gap_us [first kernel]]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
if len(sys.argv) > 3:       # only from the last launch of the kernel named here on
    last = max(i for i, e in enumerate(ev) if sys.argv[3] in e[2])
    ev = ev[last:]
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
print("kernels %d, busy %.1f ms, span %.1f ms" % (len(ev), busy / 1e6, span / 1e6))
end = ev[0][1]
gaps = []
for s, e, n in ev[1:]:
    if s > end:
        gaps.append((s - end, n))
    end = max(end, e)
gaps.sort(reverse=True)
print("idle %.1f ms in %d gaps; gaps > %.0f us:" % (sum(g for g, _ in gaps) / 1e6, len(gaps), thr))
for g, n in gaps[:40]:
    if g / 1e3 >= thr:
        print("  %8.1f us before %s" % (g / 1e3, n.replace("(anonymous namespace)::", "")[:100]))
print("gaps by size: >1 ms %d, 0.1-1 ms %d, 10-100 us %d, <10 us %d" % (sum(g >= 1e6 for g, _ in gaps), sum(1e5 <= g < 1e6 for g, _ in gaps), sum(1e4 <= g < 1e5 for g, _ in gaps), sum(g < 1e4 for g, _ in gaps)))
