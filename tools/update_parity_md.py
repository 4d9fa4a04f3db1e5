#!/usr/bin/env python3
"""Refresh PARITY.md from a GPU collection (gpurun_out/prof_final/parity_table.md = tools/parity_report.py's table, parity_prints.log = the parity
tests' printed figures): rows of the table whose first cell matches are replaced, the round's own rows are (re)written at the end of the table."""
import os, re, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "gpurun_out", "prof_final")
md = open(os.path.join(REPO, "PARITY.md")).read().splitlines()
new = []                                                # (key, row) in order: keys may repeat ("... T, ps, tracer")
for ln in open(os.path.join(src, "parity_table.md")):
    if ln.startswith("| ") and not ln.startswith("| quantity") and not ln.startswith("|---"):
        new.append((ln.split("|")[1].strip(), ln.rstrip("\n")))
out, seen = [], set()
for ln in md:
    key = ln.split("|")[1].strip() if ln.startswith("| ") and ln.count("|") >= 4 else None
    hit = next((i for i, (k, _) in enumerate(new) if k == key), None) if key else None
    if hit is not None:
        out.append(new.pop(hit)[1]); seen.add(key)
    elif "(round 6)" in ln and ln.startswith("| "):
        continue                                        # rewritten below
    else:
        out.append(ln)
prints = open(os.path.join(src, "parity_prints.log")).read()
rows = []
m = re.search(r"trip test, worst relative difference per day: \{1: ([0-9.e+-]+), 2: ([0-9.e+-]+), 3: ([0-9.e+-]+)\}", prints)
if m:
    a, b, c = (float(x) for x in m.groups())
    rows.append(f"| the reference's trip test (`trip_test_functions.py:173-189, 286-297`): T21L25 Held-Suarez, daily means of ps, ucomp, vcomp, temp, vor, div in the history file the library writes, days 1 / 2 / 3; pk, bk bit for bit (round 6) | reference daily means (`trip_T21L25.npz`) | {a:.1e} / {b:.1e} / {c:.1e} | 1e-09 / 1e-08 / 1e-08 |")
for n in (1, 10):
    m = re.search(r"developed moist T42L25 state \+ %d steps vs the reference: (\{[^}]*\}) \| response to 1 ulp in T: (\{[^}]*\})" % n, prints)
    if m:
        e, r = eval(m.group(1)), eval(m.group(2))
        f = lambda d: " / ".join(f"{d[k]:.1e}" for k in ("ug", "vg", "tg", "tr", "psg"))
        rows.append(f"| developed MOIST state (`moist_developed_T42L25`), + {n} step{'s' if n > 1 else ''}: u / v / T / q / ps, and beside it the model's own response to a one-ulp perturbation of the handed-over temperatures (round 6) | reference run continued | {f(e)}; one ulp: {f(r)} | max({'1e-11' if n == 1 else '1e-10'}, 2 x the one-ulp response) |")
# insert the round's rows behind the last table row
last = max(i for i, ln in enumerate(out) if ln.startswith("| ") and ln.count("|") >= 4)
out[last + 1:last + 1] = rows
open(os.path.join(REPO, "PARITY.md"), "w").write("\n".join(out) + "\n")
print("replaced", len(seen), "rows; added", len(rows))
