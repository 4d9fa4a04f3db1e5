#!/usr/bin/env python3
"""Summarise the memory-instruction structure of each gfx950 kernel in kernels.hip: runs of loads/stores and the
s_waitcnt vmcnt / barriers between them, plus register counts.  Used to spot serialised load phases
(waits inside a load cluster), uniform table reads left on the vector path, and spills.
usage: python tools/isa_summary.py [substring-of-kernel-name ...]"""
import re, subprocess, sys, os, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
unit = os.environ.get("ISA_UNIT", "kernels")                    # ISA_UNIT=moist: moist.hip (built like the library: -ffp-contract=off)
src = os.path.join(HERE, "..", "isca_amd", "csrc", unit + ".hip")
out = os.path.join(tempfile.gettempdir(), f"isca_{unit}.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only"] + (["-ffp-contract=off"] if unit == "moist" else []) + [src, "-o", out],
               check=True, stderr=subprocess.DEVNULL)
text = open(out).read()
meta = {m.group(1): (m.group(2), m.group(3)) for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size: (\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", text)}
for m in re.finditer(r"^(_ZN4isca[^:\s]+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if len(sys.argv) > 1 and not any(a in name for a in sys.argv[1:]):
        continue
    ev = []
    for ln in body.splitlines():
        t = ln.strip().split()
        if not t:
            continue
        op = t[0]
        if op.startswith("global_load") or op.startswith("buffer_load"): ev.append("L")
        elif op.startswith("global_store") or op.startswith("buffer_store"): ev.append("S")
        elif op.startswith("scratch_"): ev.append("X")
        elif op == "s_waitcnt" and "vmcnt" in ln: ev.append("w" + re.search(r"vmcnt\((\d+)\)", ln).group(1))
        elif op == "s_barrier": ev.append("|B|")
        elif op.startswith("ds_read") or op.startswith("ds_write"): ev.append("d")
        elif "mfma" in op: ev.append("M")
    comp, prev, n = [], None, 0
    for e in ev + [None]:
        if e == prev and e in ("L", "S", "d", "M", "X"):
            n += 1
        else:
            if prev is not None:
                comp.append(f"{prev}{n}" if prev in ("L", "S", "d", "M", "X") else prev)
            prev, n = e, 1
    sc, vg = meta.get(name, ("?", "?"))
    print(f"\n{name}\n  vgpr {vg} scratch {sc}B  lines {len(body.splitlines())}\n  " + " ".join(comp))
