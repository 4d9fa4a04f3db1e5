#!/usr/bin/env python3
"""Build recipe for oracle/_ref: the reference's own spectral-core Fortran, compiled in place.

TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or executed by the
product (isca_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may touch it, and only as the checker / reported CPU baseline.

What this does (our recipe, not the reference's mkmf/Makefile build system):
  * scans the reference source tree where it lies (/root/reference/src) for Fortran modules,
  * takes the dependency closure of the modules our harness (oracle/ref_harness.F90) uses,
  * compiles exactly those files with AMD flang (-fdefault-real-8: every real is fp64, as the
    reference's own templates do) in the reference's single-PE "nocomm" mpp mode (no -Duse_libMPI)
    and without netCDF (no -Duse_netCDF) -- unmodified and where they lie, with ONE exception for the
    moist target that is spelled out at COMPAT_EDITS below (a build-time copy with an explicit int()),
  * links the harnesses into oracle/_ref/ref_harness.x, ref_moist_harness.x and ref_shallow_harness.x.
Outputs go only to oracle/_ref/ (git-ignored; travels to the GPU box with gpurun).
No reference source is copied into this repository.

Link note: flang has no GNU `STAT` intrinsic, which mpp_io's file-size helper
(src/shared/mpp/include/mpp_io_connect.inc:870) references as an external `stat_`.  That helper is
never reached on this path (it is only used for `filesize='file'` opens), so the symbol is left
unresolved at link time (-Wl,--unresolved-symbols=ignore-all); no stand-in is written.
"""
import os, re, subprocess, sys, hashlib, json
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("ISCA_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "src")
OUT = os.path.join(HERE, "_ref")
BLD = os.path.join(OUT, "build")
FLANG = os.environ.get("FLANG", "/opt/rocm/lib/llvm/bin/flang")
CC = os.environ.get("CC", "gcc")

INCLUDES = ["shared/include", "shared/mpp/include", "shared/fms", "shared/fft",
            "shared/drifters", "shared/mpp"]
FFLAGS = ["-cpp", "-O2", "-fdefault-real-8", "-fdefault-double-8"]

# Two targets, same recipe:
#   dry   : oracle/ref_harness.F90        -> oracle/_ref/ref_harness.x        (spectral core + Held-Suarez, configs 0-2 and 4)
#   moist : oracle/ref_moist_harness.F90  -> oracle/_ref/ref_moist_harness.x  (+ the Frierson column chain of config 3:
#           idealized_moist_phys and the atmos_param / coupler modules it calls; RRTM and SOCRATES switched off with the
#           reference's own -DRRTM_NO_COMPILE / -DSOC_NO_COMPILE)
TARGETS = {
    "dry": dict(harness="ref_harness.F90", exe="ref_harness.x", build="build",
                scan=["shared", "atmos_spectral", "atmos_shared", "atmos_param/hs_forcing"],
                cppdefs=["-DINTERNAL_FILE_NML", "-DOVERLOAD_C8"]),
    "moist": dict(harness="ref_moist_harness.F90", exe="ref_moist_harness.x", build="build_moist",
                  scan=["shared", "atmos_spectral", "atmos_shared", "atmos_param", "coupler"],
                  cppdefs=["-DINTERNAL_FILE_NML", "-DOVERLOAD_C8", "-DRRTM_NO_COMPILE", "-DSOC_NO_COMPILE"],
                  # external (non-module) procedures the `use` graph cannot see: the Monin-Obukhov kernels called by monin_obukhov_mod
                  extra=["atmos_param/monin_obukhov/monin_obukhov_kernel.F90"],
                  # oracle/ref_peek.c: read access to one module-private array of the reference (t_surf), see that file's header
                  own_c=["ref_peek.c"]),
    # sibling core: src/atmos_spectral_shallow + the stirring module it shares with the barotropic core
    "shallow": dict(harness="ref_shallow_harness.F90", exe="ref_shallow_harness.x", build="build_shallow",
                    scan=["shared", "atmos_spectral/tools", "atmos_spectral/model", "atmos_shared", "atmos_spectral_shallow",
                          "atmos_spectral_barotropic"],
                    cppdefs=["-DINTERNAL_FILE_NML", "-DOVERLOAD_C8"], skip=("socrates", "rrtm_radiation", "atmos_column")),
    "barotropic": dict(harness="ref_barotropic_harness.F90", exe="ref_barotropic_harness.x", build="build_barotropic",
                       scan=["shared", "atmos_spectral/tools", "atmos_spectral/model", "atmos_shared", "atmos_spectral_barotropic"],
                       cppdefs=["-DINTERNAL_FILE_NML", "-DOVERLOAD_C8"], skip=("socrates", "rrtm_radiation", "atmos_column")),
    # topog_regularization_mod (ocean_topog_smoothing /= 0) driven through its two public routines: oracle/ref_topog_harness.F90
    "topog": dict(harness="ref_topog_harness.F90", exe="ref_topog_harness.x", build="build_topog",
                  scan=["shared", "atmos_spectral", "atmos_shared", "atmos_param/hs_forcing"],
                  cppdefs=["-DINTERNAL_FILE_NML", "-DOVERLOAD_C8"]),
    # The module-level drop-in (bindings/fortran/dropin): THIS REPOSITORY's modules under the reference's names (spectral_dynamics_mod,
    # transforms_mod, press_and_geopot_mod, hs_forcing_mod, ...) in front of isca_amd/lib/libisca_dyn.so, compiled together with the
    # reference's own infrastructure modules (fms_mod, time_manager_mod, tracer_manager_mod, ... from src/shared, in place) and linked with
    # the SAME oracle/ref_harness.F90 as the "dry" target: the harness then drives the GPU library through the reference's interface.
    "dropin": dict(harness="ref_harness.F90", exe="ref_harness_gpu.x", build="build_dropin", scan=["shared"],
                   own=[os.path.join(os.path.dirname(HERE), "bindings", "fortran", "isca_dyn_c.F90"),
                        os.path.join(os.path.dirname(HERE), "bindings", "fortran", "dropin")],
                   cppdefs=["-DINTERNAL_FILE_NML", "-DOVERLOAD_C8"],
                   link=["-L" + os.path.join(os.path.dirname(HERE), "isca_amd", "lib"), "-lisca_dyn", "-Wl,-rpath,$ORIGIN/../../isca_amd/lib"]),
    # ... and the main program's side of it: bindings/fortran/dropin/drive_atmos_model.F90 (atmos_model's time loop) on this
    # repository's atmosphere_mod
    "dropin_atmos": dict(harness=os.path.join(os.path.dirname(HERE), "bindings", "fortran", "dropin", "drive_atmos_model.F90"),
                         exe="drive_atmos_model_gpu.x", build="build_dropin_atmos", scan=["shared"],
                         own=[os.path.join(os.path.dirname(HERE), "bindings", "fortran", "isca_dyn_c.F90"),
                              os.path.join(os.path.dirname(HERE), "bindings", "fortran", "dropin")],
                         cppdefs=["-DINTERNAL_FILE_NML", "-DOVERLOAD_C8"],
                         link=["-L" + os.path.join(os.path.dirname(HERE), "isca_amd", "lib"), "-lisca_dyn", "-Wl,-rpath,$ORIGIN/../../isca_amd/lib"]),
}
# One compiler-compatibility edit, applied to a BUILD-TIME COPY under oracle/_ref/<build>/compat/ (git-ignored, never in
# this repository): qe_moist_convection.F90 indexes lcl_temp_table with a variable declared `real` (get_lcl_temp, :1060,
# :1080).  gfortran/ifort convert such a subscript to integer silently; flang rejects it.  The copy spells out the same
# conversion, int(iv_floor) -- iv_floor holds floor(...)+1, so the value is unchanged.  Every other file is compiled
# where it lies, unmodified.
COMPAT_EDITS = {
    "atmos_param/qe_moist_convection/qe_moist_convection.F90": [
        ("lcl_temp_table(iv_floor+1)*w_ceil - lcl_temp_table(iv_floor)*(w_ceil-1)",
         "lcl_temp_table(int(iv_floor)+1)*w_ceil - lcl_temp_table(int(iv_floor))*(w_ceil-1)"),
    ],
}
SKIP_DIR_WORDS = ("atmos_spectral_barotropic", "atmos_spectral_shallow", "socrates", "rrtm_radiation", "atmos_column")

mod_re = re.compile(r"^\s*module\s+(\w+)\s*$", re.I)
use_re = re.compile(r"^\s*use\s*(?:,\s*\w+\s*::)?\s*(\w+)", re.I)
prog_re = re.compile(r"^\s*program\s+\w+", re.I)


def scan(path):
    mods, uses, is_prog = set(), set(), False
    depth_if = []  # crude: skip `#ifdef test_*` blocks that hold print-only test programs
    with open(path, errors="replace") as f:
        for line in f:
            s = line.strip()
            if s.startswith("#"):
                m = re.match(r"#\s*ifdef\s+(\w+)", s)
                if m:
                    depth_if.append(m.group(1).lower().startswith("test_"))
                elif re.match(r"#\s*if", s):
                    depth_if.append(False)
                elif re.match(r"#\s*endif", s) and depth_if:
                    depth_if.pop()
                continue
            if any(depth_if):
                continue
            m = mod_re.match(line)
            if m and m.group(1).lower() != "procedure":
                mods.add(m.group(1).lower())
            m = use_re.match(line)
            if m:
                uses.add(m.group(1).lower())
            if prog_re.match(line):
                is_prog = True
    return mods, uses, is_prog


def build_target(name):
    t = TARGETS[name]
    BLD = os.path.join(OUT, t["build"])
    os.makedirs(BLD, exist_ok=True)
    files = {}
    for d in t["scan"]:
        for root, _, names in os.walk(os.path.join(SRC, d)):
            if any(w in root.lower() for w in t.get("skip", SKIP_DIR_WORDS)):
                continue
            for n in names:
                if n.endswith((".F90", ".f90")) and not n.startswith("test_"):
                    p = os.path.join(root, n)
                    files[p] = scan(p)
    mod2file = {}
    for src in t.get("own", []):          # this repository's modules first: they take the reference's module names
        own_files = [src] if os.path.isfile(src) else [os.path.join(src, n) for n in sorted(os.listdir(src)) if n.endswith(".F90")]
        for p in own_files:
            if p == os.path.join(HERE, t["harness"]):
                continue
            files[p] = scan(p)
            for m in files[p][0]:
                mod2file[m] = p
    for p, (mods, _, is_prog) in files.items():
        if is_prog and not mods:
            continue
        for m in mods:
            mod2file.setdefault(m, p)
    harness = os.path.join(HERE, t["harness"])
    hm, hu, _ = scan(harness)
    order, seen = [], set()

    def visit(p, uses):
        for u in sorted(uses):
            q = mod2file.get(u)
            if q and q not in seen:
                seen.add(q)
                visit(q, files[q][1])
                order.append(q)

    visit(harness, hu)
    for rel in t.get("extra", []):
        q = os.path.join(SRC, rel)
        if q not in seen:
            seen.add(q)
            visit(q, files[q][1])
            order.append(q)
    inc = sum((["-I", os.path.join(SRC, i)] for i in INCLUDES), []) + ["-I", BLD]
    stamp_path = os.path.join(BLD, "stamps.json")
    stamps = json.load(open(stamp_path)) if os.path.exists(stamp_path) else {}
    objs = []
    rebuilt_any = False
    for p in order:
        o = os.path.join(BLD, os.path.basename(p).rsplit(".", 1)[0] + ".o")
        objs.append(o)
        h = hashlib.sha1(open(p, "rb").read()).hexdigest()
        if os.path.exists(o) and stamps.get(p) == h and not rebuilt_any:
            continue
        src = p
        rel = os.path.relpath(p, SRC) if p.startswith(SRC) else os.path.relpath(p, os.path.dirname(HERE))
        if rel in COMPAT_EDITS:
            text = open(p, errors="replace").read()
            for old, new in COMPAT_EDITS[rel]:
                if text.count(old) != 1:
                    print("compat edit does not apply to", rel); return 1
                text = text.replace(old, new)
            os.makedirs(os.path.join(BLD, "compat"), exist_ok=True)
            src = os.path.join(BLD, "compat", os.path.basename(p))
            open(src, "w").write(text)
            print("compat copy:", rel, "->", os.path.relpath(src, HERE))
        cmd = [FLANG] + FFLAGS + t["cppdefs"] + inc + ["-I", os.path.dirname(p), "-module-dir", BLD, "-c", src, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print("FAILED:", " ".join(cmd)); print(r.stderr[-4000:]); return 1
        stamps[p] = h; rebuilt_any = True
        json.dump(stamps, open(stamp_path, "w"))
        print("compiled", rel, flush=True)
    # the reference's few C helpers used by mpp/memutils (compiled in place, unmodified)
    cobjs = []
    for c in ["shared/mpp/nsclock.c", "shared/mpp/threadloc.c", "shared/mpp/affinity.c",
              "shared/memutils/memuse.c"]:
        p = os.path.join(SRC, c)
        o = os.path.join(BLD, os.path.basename(c)[:-2] + "_c.o")
        if not os.path.exists(o):
            r = subprocess.run([CC, "-O2", "-D__IFC", "-c", p, "-o", o], capture_output=True, text=True)
            if r.returncode != 0:
                print("C helper failed (skipped):", c, r.stderr[-500:])
                continue
        cobjs.append(o)
    for c in t.get("own_c", []):          # this repository's C helpers of the harness (oracle/*.c)
        o = os.path.join(BLD, c[:-2] + "_own_c.o")
        r = subprocess.run([CC, "-O2", "-c", os.path.join(HERE, c), "-o", o], capture_output=True, text=True)
        if r.returncode != 0:
            print("FAILED:", c, r.stderr[-2000:]); return 1
        cobjs.append(o)
    ho = os.path.join(BLD, os.path.basename(t["harness"])[:-4] + ".o")
    cmd = [FLANG] + FFLAGS + t["cppdefs"] + inc + ["-module-dir", BLD, "-c", harness, "-o", ho]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        print("FAILED harness:"); print(r.stderr[-6000:]); return 1
    exe = os.path.join(OUT, t["exe"])
    cmd = [FLANG, "-o", exe, ho] + objs + cobjs + t.get("link", []) + ["-Wl,--unresolved-symbols=ignore-all"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        print("FAILED link:"); print(r.stderr[-6000:]); return 1
    nown = sum(1 for p in order if not p.startswith(SRC))
    print("built", exe, "from", len(order) - nown, "reference Fortran files" + (" and %d of this repository's" % nown if nown else ""))
    return 0


def main():
    if not os.path.isdir(SRC):
        print("build_ref: reference tree not present (%s); keeping prebuilt oracle/_ref" % SRC)
        return 0
    want = [a for a in sys.argv[1:] if a in TARGETS] or ["dry"] + (["moist"] if os.path.exists(os.path.join(HERE, TARGETS["moist"]["harness"])) else [])
    for name in want:
        rc = build_target(name)
        if rc:
            return rc
    return 0


if __name__ == "__main__":
    sys.exit(main())
