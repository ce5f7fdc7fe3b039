#!/usr/bin/env python3
"""Diagnostic: from a rocprofv3 kernel trace (…_kernel_trace.csv), how busy was the GPU -- union of all kernels, idle gaps, and per
kernel kind: count, mean duration, mean start-to-start interval.  usage: trace_gaps.py <kernel_trace.csv> [skip fraction]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:], r.get("Queue_Id", "")) for r in rows))
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo = t0 + (t1 - t0) * skip                       # steady state: the last part of the run
ev = [e for e in ev if e[0] >= lo]
span = max(e[1] for e in ev) - ev[0][0]
busy, idle_gaps, cur_end = 0, [], ev[0][0]
for s, e, _, _ in ev:
    if s > cur_end:
        idle_gaps.append(s - cur_end)
        cur_end = s
    if e > cur_end:
        busy += e - cur_end
        cur_end = e
print("window %.1f ms, some kernel running %.1f %% of it; %d idle gaps, total %.1f %%, longest %.1f us, gaps > 5 us: %d" %
      (span / 1e6, 100 * busy / span, len(idle_gaps), 100 * sum(idle_gaps) / span, max(idle_gaps or [0]) / 1e3, sum(g > 5000 for g in idle_gaps)))
kinds = {}
for s, e, k, q in ev:
    kinds.setdefault(k, []).append((s, e, q))
for k, v in sorted(kinds.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    d = [e - s for s, e, _ in v]
    st = [b[0] - a[0] for a, b in zip(v, v[1:])]
    qs = sorted(set(q for _, _, q in v))
    print("  %-44s n %5d  mean %8.1f us  busy %5.1f %% of the window  start-to-start %8.1f us  queues %s" %
          (k, len(v), sum(d) / len(d) / 1e3, 100 * sum(d) / span, (sum(st) / len(st) / 1e3) if st else 0, ",".join(qs)))
# per queue: fraction of the window with a kernel of that queue running
qb = {}
for s, e, k, q in ev:
    qb.setdefault(q, []).append((s, e))
for q, v in sorted(qb.items()):
    b, ce = 0, v[0][0]
    for s, e in sorted(v):
        if s > ce:
            ce = s
        if e > ce:
            b += e - ce
            ce = e
    print("  queue %-6s busy %5.1f %% of the window (%d kernels)" % (q, 100 * b / span, len(v)))
# a stretch of the timeline in the middle of the window: start (us from the stretch's first kernel), duration, queue, kernel
if len(sys.argv) > 3:
    mid = len(ev) // 2
    base = ev[mid][0]
    for s, e, k, q in ev[mid:mid + int(sys.argv[3])]:
        print("  t %9.1f us  dur %7.1f us  queue %-3s %s" % ((s - base) / 1e3, (e - s) / 1e3, q, k.split("::")[-1][:40]))
