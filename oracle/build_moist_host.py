#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY: host build (g++) of the moist column routines for the CPU tests -> oracle/_ref/libmoist_host.so"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT, "libmoist_host.so")


def build():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(HERE, "moist_host.cpp")
    deps = [src] + [os.path.join(HERE, "..", "isca_amd", "csrc", f) for f in ("moist_physics.h", "moist_tables.h")]
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) > os.path.getmtime(d) for d in deps):
        return LIB
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", LIB, src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("moist host build failed:\n" + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build())
