"""Build the in-tree HIP shared library isca_amd/lib/libisca_dyn.so for gfx950 (hipcc cross-compiles
without a GPU).  Used by __graft_entry__.build() and importable on its own: python -m isca_amd.build"""
import os, subprocess, sys, hashlib, json

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libisca_dyn.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

UNITS = [  # (source, extra flags)
    ("tables.cpp", ["-ffp-contract=off"]),      # host tables: reproduce the reference's non-FMA fp64 results
    ("comm.cpp", []),                           # RCCL bound with dlopen (no link-time dependency)
    ("comm_peer.hip", []),                      # device-resident exchange between the ranks of one node: stores into hipIpc-mapped peer buffers, one kernel per exchange
    ("comm_ipc.cpp", []),                       # host-staged exchange for ranks sharing one GPU (verification of the sharded C++ loop)
    ("restart_nc.cpp", []),                     # restart files in the netCDF classic format (no netCDF library)
    ("history_nc.cpp", []),                     # diag_table + history files (diag_manager's part for the fields the device accumulates)
    ("topog.cpp", ["-ffp-contract=off"]),        # get_topography for a handed-over height field: truncation or regularisation over the ocean (host arithmetic)
    ("kernels.hip", []),
    ("legendre.hip", []),
    ("moist.hip", ["-ffp-contract=off"]),       # moist column physics: no contraction, like the reference build (regime tests)
    ("api.hip", []),
]
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def _digest(paths):
    h = hashlib.sha1()
    for p in sorted(paths):
        h.update(open(p, "rb").read())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    inc = os.path.join(HERE, "..", "include")
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    stamp = os.path.join(LIBDIR, "build_stamp.json")
    dig = _digest(srcs)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and json.load(open(stamp)).get("digest") == dig:
        return LIB
    objs = []
    procs = []
    # per-unit digests (the unit's source + every header / .inc + its flags): only what changed is recompiled
    old_units = json.load(open(stamp)).get("units", {}) if os.path.exists(stamp) and not force else {}
    shared = [p for p in srcs if p.endswith((".h", ".inc"))]
    units = {}
    for src, extra in UNITS:
        obj = os.path.join(LIBDIR, src.rsplit(".", 1)[0] + ".o")
        units[src] = hashlib.sha1((_digest(shared + [os.path.join(CSRC, src)]) + " ".join(COMMON + extra)).encode()).hexdigest()
        objs.append(obj)
        if os.path.exists(obj) and old_units.get(src) == units[src]:
            continue
        cmd = [HIPCC] + COMMON + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + ["-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    json.dump({"digest": dig, "units": units}, open(stamp, "w"))
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
