#!/usr/bin/env python3
"""Analyse a scheduling trace of the Legendre kernels (library built with -DLEG_TRACE, see legendre.hip):
usage: python tools/dev/leg_trace.py gpurun_out/leg_trace_fwd.bin"""
import sys
import numpy as np
rec = np.dtype([("t0", "<i8"), ("t1", "<i8"), ("ta", "<i8"), ("tb", "<i8"), ("hw", "<u4"), ("xcc", "<u4"), ("ml", "<u4"), ("what", "<u4")])
r = np.fromfile(sys.argv[1], dtype=rec)
r = r[r["t1"] > 0]
t0 = r["t0"].min()
s = (r["t0"] - t0) * 0.01; e = (r["t1"] - t0) * 0.01           # microseconds
simd = (r["hw"] >> 4) & 3; cu = (r["hw"] >> 8) & 15; sh = (r["hw"] >> 12) & 1; se = (r["hw"] >> 13) & 7; xcc = r["xcc"] & 15
key = ((xcc * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
print("working waves", len(r), "kernel span %.1f us" % e.max(), "distinct SIMDs", len(np.unique(key)), "XCCs", np.unique(xcc))
dur = e - s
print("wave duration us: min %.1f median %.1f max %.1f ; sum %.0f us ; per 1024 SIMDs %.1f us" % (dur.min(), np.median(dur), dur.max(), dur.sum(), dur.sum() / 1024))
# occupancy over time
ts = np.linspace(0, e.max(), 41)
print("time(us) : resident working waves")
for a, b in zip(ts[:-1], ts[1:]):
    mid = 0.5 * (a + b)
    print("  %6.1f : %5d" % (mid, int(((s <= mid) & (e > mid)).sum())))
# per-SIMD busy (union of intervals) and end time
busy = []; ends = []
for k in np.unique(key):
    m = key == k
    iv = sorted(zip(s[m], e[m]))
    tot = 0.0; cur_s, cur_e = iv[0]
    for a, b in iv[1:]:
        if a > cur_e: tot += cur_e - cur_s; cur_s, cur_e = a, b
        else: cur_e = max(cur_e, b)
    tot += cur_e - cur_s
    busy.append(tot); ends.append(max(b for _, b in iv))
busy = np.array(busy); ends = np.array(ends)
print("per-SIMD time with >=1 wave: mean %.1f min %.1f max %.1f us ; last wave ends: mean %.1f min %.1f max %.1f" %
      (busy.mean(), busy.min(), busy.max(), ends.mean(), ends.min(), ends.max()))
# duration by item kind
for w in np.unique(r["what"]):
    m = r["what"] == w
    ta = (r["ta"][m] - r["t0"][m]) * 0.01; tb = (r["tb"][m] - r["t0"][m]) * 0.01
    ta = ta[r["ta"][m] > 0]
    print("  what=%3d waves %5d mean dur %.1f us  start mean %.1f | first k-steps done after %.1f, last MFMA issued after %.1f" %
          (w, m.sum(), dur[m].mean(), s[m].mean(), ta.mean() if ta.size else float("nan"), tb.mean()))
