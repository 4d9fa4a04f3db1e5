#!/usr/bin/env python3
"""DESIGN.md section 6 from the committed round summaries (profiles/<tag>_*): rewrites the block between the R_NUMBERS markers.
usage: python tools/design_numbers.py [tag]"""
import csv, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = lambda name: os.path.join(REPO, "profiles", f"{tag}_{name}")
b = json.load(open(P("T85L40_bench.json")))
def stats(name):
    return {r["kernel"].strip('"'): float(r["avg_us"]) for r in csv.DictReader(open(P(name)))}
def traffic(name):
    return json.load(open(P(name)))["bytes_per_launch"]
out = []
out.append(f"Headline (`profiles/{tag}_T85L40_bench.json`, the plain `python bench.py` line of the final build): **{b['ms_per_step']:.4f} ms per step = "
           f"{b['value']:.0f} simulated years per day** on one MI355X (500 timed steps; round 5: 0.1775 ms = 4 694); through the Fortran drop-in's `atmos_model` loop "
           f"{b['dropin_ms_per_step']:.4f} ms.  The reference on one host core of the same box in the same run: {b['cpu_baseline']['ms_per_step']:.1f} ms per step = "
           f"{b['cpu_baseline']['value']:.2f} simulated years per day (8 concurrent copies: {b['cpu_baseline']['multi_core']['value']:.1f}).  "
           f"Step against the HBM roofline: {b['step_bytes']['algorithmic_bytes'] / 1e6:.0f} MB algorithmic / {b['ms_per_step']:.4f} ms = "
           f"{b['step_roofline']['achieved'] / 1e3:.2f} TB/s = {b['step_roofline']['frac']:.2f} of 8 TB/s.\n")
for wl, (I, J, N, L) in (("T85L40", (256, 128, 85, 40)), ("T170L60", (512, 256, 170, 60))):
    st = stats(f"{wl}_kernel_stats.csv")
    tr = traffic("pmc_traffic.json" if wl == "T85L40" else f"{wl}_pmc_traffic.json")
    bj = json.load(open(P(f"{wl}_bench_under_rocprof.json")))
    fi = bj["step_bytes"]["fourier_intermediate_bytes"]
    nlf_inv = int(round(fi / (2.0 * 16.0 * (N + 1) * J))) - (4 * L + 1)
    alg = bench.algorithmic_bytes(I, J, N + 1, N, L, nlf_inv=nlf_inv)
    flt = J * (N + 1) * (N + 4)
    out.append(f"**{wl}** (rocprofv3 `--kernel-trace --stats` averages inside the step, `profiles/{tag}_{wl}_kernel_stats.csv`; HBM bytes per launch from the separate "
               f"`--pmc` passes, 2·FETCH_SIZE + WRITE_SIZE, `profiles/{tag}_{'pmc_traffic.json' if wl == 'T85L40' else wl + '_pmc_traffic.json'}`; step under the profiler "
               f"{bj['ms_per_step']:.4f} ms):\n")
    out.append("| kernel | µs | algorithmic MB | → TB/s | of 8 TB/s | counted MB | counted / algorithmic | FP64 MFMA TFLOP/s (of 78.6) |")
    out.append("|---|---|---|---|---|---|---|---|")
    tot_a = tot_c = 0.0
    for timer, kname in (("column", "k_column_sig"), ("tracer_horiz", "k_tracer_horiz"), ("tracer_vert", "k_tracer_vert"), ("fft_fwd", "k_fft_fwd3"),
                         ("legendre_fwd", "k_leg_fwd"), ("spec_update", "k_spec_update"), ("legendre_inv", "k_leg_inv_coop:fused"), ("fft_inv", "k_fft_inv3"),
                         ("fixer_sums", "k_fixer_sums"), ("fixer_finish", "k_fixer_finish")):
        if kname not in st:
            continue
        us, a, c = st[kname], alg.get(timer), tr.get(kname)
        tf = ""
        if timer.startswith("legendre"):
            fl = flt * ((4 * L + 1) if timer == "legendre_fwd" else nlf_inv)
            tf = f"{fl / us / 1e6:.1f} ({fl / us / 1e6 / 78.6:.2f})"
        if a:
            tot_a += a
        if c:
            tot_c += c
        out.append(f"| `{kname}` | {us:.1f} | {a / 1e6:.1f} | {a / us / 1e6:.2f} | {a / us / 1e6 / 8:.2f} | {c / 1e6:.1f} | {c / a:.2f} | {tf} |" if a and c else
                   f"| `{kname}` | {us:.1f} | — | — | — | {(c or 0) / 1e6:.1f} | — | |")
    out.append(f"| sum of the kernels (the Fourier intermediate counts on both sides of it) | | {tot_a / 1e6:.0f} | | | {tot_c / 1e6:.0f} | {tot_c / tot_a:.2f} | |\n")
sc = json.load(open(P("shard_compute.json")))
out.append("**The sharded step's compute on one GPU** (`bench.py: shard_compute_ms`, P processes taking turns, slowest rank, exchanges excluded; "
           f"`profiles/{tag}_shard_compute.json` is the collection under rocprofv3, whose events are a little longer; `profiles/{tag}_<workload>_P<n>_kernel_stats.csv` "
           "rank 0's kernels):\n")
out.append("| | one GPU | P = 2 | P = 4 | P = 8 |")
out.append("|---|---|---|---|---|")
for wl in ("T85L40", "T170L60"):
    v = b["shard_compute_ms"][wl]
    one = b["ms_per_step"] if wl == "T85L40" else next(x["ms_per_step"] for k, x in b["other_workloads"].items() if k.startswith("T170L60"))
    cell = lambda P_: (f"{v[f'P={P_}']['segments_ms']:.3f} (kernels {v[f'P={P_}']['main_stream_ms']:.3f}, side {v[f'P={P_}']['side_stream_ms']:.3f})" if f"P={P_}" in v and "segments_ms" in v[f"P={P_}"] else "—")
    out.append(f"| {wl}: `segments_ms` (per-kernel event sum, side stream), ms | {one:.3f} (whole step) | {cell(2)} | {cell(4)} | {cell(8)} |")
out.append("\n(`k_leg_fwd` at T170L60 counts 43 MB for 226 MB of Fourier rows it must read: its rows were written by the kernel before and FETCH_SIZE does not see what the 256 MB memory-side cache serves; the same holds, less visibly, for every consumer that follows its producer.)\n")
for k, x in b["other_workloads"].items():
    if "ms_per_step" in x and "roofline" in x and x["roofline"]:
        r = x["roofline"]
        out.append(f"* {k}: **{x['ms_per_step']:.4f} ms per step** = {x.get('sim_years/day', 0):.0f} simulated years per day; dominant kernel `{r['kernel']}` "
                   f"{1e3 * r['avg_launch_ms']:.1f} µs by HIP events, {r['frac']:.2f} of the {'HBM' if r['bound'] == 'hbm' else 'MFMA'} peak.")
    elif "ms_per_step" in x:
        out.append(f"* {k}: {x['ms_per_step']:.4f} ms per step.")
block = "\n".join(out)
d = open(os.path.join(REPO, "DESIGN.md")).read()
B, E = "<!-- R_NUMBERS_BEGIN -->", "<!-- R_NUMBERS_END -->"
if B in d:
    d = d[:d.index(B) + len(B)] + "\n" + block + "\n" + d[d.index(E):]
else:
    d = d.replace("ROUND6_NUMBERS", B + "\n" + block + "\n" + E)
open(os.path.join(REPO, "DESIGN.md"), "w").write(d)
print(block[:3000])
