"""The reference is Fortran: bindings/fortran/isca_dyn_c.F90 is the bind(C) module a maintainer adds to it, and
drive_held_suarez.F90 the calls a replacement atmosphere_mod makes.  Compiled here with the image's flang against the in-tree library
and run on the GPU: 144 Held-Suarez steps at T21L25 from Fortran must land on the reference run (tests/golden/run_T21L25.npz)."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLANG = os.environ.get("FLANG", "/opt/rocm/lib/llvm/bin/flang")


def test_fortran_driver_matches_reference_run(tmp_path, golden_dir):
    if not os.path.exists(FLANG):
        pytest.skip("no flang in this image")
    from isca_amd import build
    build.build(verbose=False)
    src = os.path.join(REPO, "bindings", "fortran")
    lib = os.path.join(REPO, "isca_amd", "lib")
    mod_o = str(tmp_path / "isca_dyn_c.o")
    exe = str(tmp_path / "drive_held_suarez.x")
    subprocess.run([FLANG, "-c", os.path.join(src, "isca_dyn_c.F90"), "-o", mod_o, "-module-dir", str(tmp_path)], check=True, capture_output=True)
    subprocess.run([FLANG, os.path.join(src, "drive_held_suarez.F90"), mod_o, "-I", str(tmp_path), "-L", lib, "-lisca_dyn",
                    "-Wl,-rpath," + lib, "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    nums = lambda tag: [float(x) for x in re.search(tag + r"\s*(.*)", out).group(1).split()]
    g = np.load(os.path.join(golden_dir, "run_T21L25.npz"))
    tg, ug, ps = g["st_tg_000144"], g["st_ug_000144"], g["st_psg_000144"]
    tmin, tmax, umax = nums("FORTRAN_STATE Tmin,Tmax,maxabsU=")
    assert abs(tmin - tg.min()) < 1e-9 and abs(tmax - tg.max()) < 1e-9 and abs(umax - np.abs(ug).max()) < 1e-9
    t_pt, u_pt = nums(r"FORTRAN_POINT tg\(5,7,20\),ug\(33,12,3\)=")
    assert abs(t_pt - tg[19, 6, 4]) < 1e-9 and abs(u_pt - ug[2, 11, 32]) < 1e-9          # Fortran (lon, lat, lev) = numpy [lev, lat, lon]
    from oracle.isca_oracle import Config, SpectralCore
    sc = SpectralCore(Config.resolution("T21", 25))
    (mean_ps,) = nums("FORTRAN_MEAN_PS")
    assert abs(mean_ps - sc.area_weighted_global_mean(ps)) < 1e-6
    re_, im_ = nums(r"FORTRAN_SPEC ts\(0,0,25\)")
    ts = sc.trans_grid_to_spherical(tg)
    assert abs(re_ - ts[24, 0, 0].real) < 1e-9 and abs(im_) < 1e-12
    assert "FORTRAN_ERROR" in out and "unknown field" in out          # the FATAL convention: non-zero return + message


def test_fortran_external_physics_driver(tmp_path, golden_dir):
    """The physics / dynamics seam of atmosphere.F90:300-329 from Fortran (physics = 2): the driver evaluates hs_forcing itself (through
    isca_hs_forcing / isca_hs_tracer_source_sink on the fields the library hands out) and feeds the tendencies to isca_dyn_dynamics
    (= spectral_dynamics, spectral_dynamics.F90:780-795); 144 steps land on the reference run."""
    if not os.path.exists(FLANG):
        pytest.skip("no flang in this image")
    src = os.path.join(REPO, "bindings", "fortran")
    lib = os.path.join(REPO, "isca_amd", "lib")
    mod_o, exe = str(tmp_path / "isca_dyn_c.o"), str(tmp_path / "drive_external_physics.x")
    subprocess.run([FLANG, "-c", os.path.join(src, "isca_dyn_c.F90"), "-o", mod_o, "-module-dir", str(tmp_path)], check=True, capture_output=True)
    subprocess.run([FLANG, os.path.join(src, "drive_external_physics.F90"), mod_o, "-I", str(tmp_path), "-L", lib, "-lisca_dyn",
                    "-Wl,-rpath," + lib, "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    nums = lambda tag: [float(x) for x in re.search(tag + r"\s*(.*)", r.stdout).group(1).split()]
    g = np.load(os.path.join(golden_dir, "run_T21L25.npz"))
    tg, ug, tr = g["st_tg_000144"], g["st_ug_000144"], g["st_tr1_000144"]
    tmin, tmax, umax = nums("FORTRAN_STATE Tmin,Tmax,maxabsU=")
    assert abs(tmin - tg.min()) < 1e-9 and abs(tmax - tg.max()) < 1e-9 and abs(umax - np.abs(ug).max()) < 1e-9
    t_pt, u_pt, r_pt = nums(r"FORTRAN_POINT tg\(5,7,20\),ug\(33,12,3\),tr\(9,30,25\)=")
    assert abs(t_pt - tg[19, 6, 4]) < 1e-9 and abs(u_pt - ug[2, 11, 32]) < 1e-9 and abs(r_pt - tr[24, 29, 8]) < 1e-9 * np.abs(tr).max()
    assert "FORTRAN_ERROR" in r.stdout and "physics = 2" in r.stdout


def test_fortran_moist_driver(tmp_path, golden_dir):
    """The Frierson configuration set from Fortran through the nested bind(C) types (isca_moist_config, bk array): 144 steps on the GPU land
    on the reference's moist run."""
    if not os.path.exists(FLANG):
        pytest.skip("no flang in this image")
    src = os.path.join(REPO, "bindings", "fortran")
    lib = os.path.join(REPO, "isca_amd", "lib")
    mod_o, exe = str(tmp_path / "isca_dyn_c.o"), str(tmp_path / "drive_frierson.x")
    subprocess.run([FLANG, "-c", os.path.join(src, "isca_dyn_c.F90"), "-o", mod_o, "-module-dir", str(tmp_path)], check=True, capture_output=True)
    subprocess.run([FLANG, os.path.join(src, "drive_frierson.F90"), mod_o, "-I", str(tmp_path), "-L", lib, "-lisca_dyn", "-Wl,-rpath," + lib,
                    "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    g = np.load(os.path.join(golden_dir, "moist_run_T21L25.npz"))
    tg, q = g["st_tg_000144"], g["st_q_000144"]
    tmin, tmax, qmax, qpt = [float(x) for x in re.search(r"FORTRAN_MOIST Tmin,Tmax,qmax,q\(10,16,25\)=\s*(.*)", r.stdout).group(1).split()]
    assert abs(tmin - tg.min()) < 1e-7 and abs(tmax - tg.max()) < 1e-7 and abs(qmax - q.max()) < 1e-10 and abs(qpt - q[24, 15, 9]) < 1e-10
    smin, smax = [float(x) for x in re.search(r"FORTRAN_TSURF min,max=\s*(.*)", r.stdout).group(1).split()]
    assert 230.0 < smin < smax < 310.0
