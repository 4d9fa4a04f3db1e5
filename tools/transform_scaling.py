#!/usr/bin/env python3
"""Per-kernel times of one (s2g, g2s) transform pair vs batch size (level-fields), data resident in HBM."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isca_amd import dyncore
for res, L in (("T85", 40), ("T170", 60)):
    dc = dyncore.DynCore(dyncore.default_config(res, num_levels=L))
    for nf in (8, 32, 64, 161, 283, 7 * L + 3):
        if nf > 7 * L + 3: continue
        pair, k = dc.bench_transform_pair(nf, 30)
        print(res, "nfields", nf, "pair_us", round(1e3 * pair, 1), {a: round(1e3 * b, 1) for a, b in k.items()}, flush=True)
    dc.close()
