"""The C++ host-side mirror of the reference class surface (include/StaticFusionCompat.hpp):
compiles and links against libsf_hip.so on CPU; on the GPU box the reference drivers' frame loop
written against it reproduces the oracle's pose."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, driver_params, make_solver

SRC = os.path.join(ROOT, "tests", "cpp", "compat_driver.cpp")
LIBDIR = os.path.join(ROOT, "staticfusion_amd", "csrc")


def build(tmp_path):
    exe = str(tmp_path / "compat_driver")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L" + LIBDIR, "-lsf_hip", "-Wl,-rpath," + LIBDIR, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_compat_header_compiles_and_links(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "libsf_hip.so")):
        subprocess.check_call(["make", "-C", LIBDIR])
    assert os.path.exists(build(tmp_path))


@pytest.mark.gpu
def test_compat_driver_matches_oracle(tmp_path, ora, pair):
    from staticfusion_amd.synth import pose_delta

    exe = build(tmp_path)
    pr = pair(seed=1234, sphere=True, rows=240, cols=320)
    blob = tmp_path / "pair.bin"
    with open(blob, "wb") as f:
        for img in (pr["old"][0], pr["old"][1], pr["new"][0], pr["new"][1]):
            f.write(np.ascontiguousarray(img.T, dtype=np.float32).tobytes())  # column-major
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools", "golden"))
    from make_golden_input import synth_frame

    color, depth = synth_frame(480, 640, 77)
    frame = tmp_path / "frame.bin"
    with open(frame, "wb") as f:
        f.write(color.tobytes())
        f.write(depth.tobytes())
    out = subprocess.check_output([exe, str(blob), str(frame)]).decode().split("\n")
    T = np.array([[float(x) for x in out[r].split()] for r in range(4)])
    dyn = float(out[4].split()[1])
    s = make_solver(ora, 240, 320, driver_params(ora), pr)
    s.build_pyramid(True)
    s.run_solver(True)
    s.build_segm_image()
    rot, trans = pose_delta(s.T(), T)
    assert rot <= 1e-4 and trans <= 1e-4
    assert dyn == pytest.approx(float((s.b_image() < 0.5).mean()), abs=1e-6)
    # loadImageFromDecoded + getFilteredDepth through the class surface: the same images as the oracle's
    so = make_solver(ora, 240, 320, driver_params(ora))
    so.load_frame(0, color, depth, 2)
    so.filter_depth()
    d, i = so.current()
    got = out[5].split()
    assert got[0] == "input_stage"
    assert float(got[1]) == pytest.approx(float(d.astype(np.float64).sum()), abs=1e-6)
    assert float(got[2]) == pytest.approx(float(i.astype(np.float64).sum()), abs=1e-6)
    assert int(got[3]) == int(so.input_image(0).astype(np.uint64).sum())


def build_c_example(tmp_path):
    exe = str(tmp_path / "batch_throughput")
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "batch_throughput.c"),
                           "-o", exe, "-L" + LIBDIR, "-lsf_hip", "-Wl,-rpath," + LIBDIR, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lm"])
    return exe


def test_c_example_compiles_against_the_header(tmp_path):
    """include/sf.h is a C header: examples/batch_throughput.c builds with a C compiler and links the library"""
    assert os.path.exists(build_c_example(tmp_path))


@pytest.mark.gpu
def test_c_example_runs(tmp_path):
    out = subprocess.check_output([build_c_example(tmp_path), "64", "3"]).decode()
    assert "hip:gfx950" in out and "frames/s" in out
