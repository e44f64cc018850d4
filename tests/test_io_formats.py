"""On-disk formats either side of the path (include/sf_io.h, libsf_io.so):
association file (reference FrontEnd.cpp:183-214), PNG frames as cv::imread delivers them to
FrontEnd.cpp:220,240, trajectory lines (Utils/Datasets.cpp:252-265, Reconstruction.cpp:53-81).
The C++ library against (i) the source arrays the test's own PNG encoder started from and
(ii) the independent Python restatement oracle/io_oracle.py. Host-only: no GPU needed."""
import os

import numpy as np
import pytest

from png_util import encode_png
from oracle import io_oracle
from staticfusion_amd import io as sfio


@pytest.fixture(scope="module")
def lib():
    return sfio.Io()  # no fallback: a missing libsf_io.so is an error


CASES = [  # (colour type, bit depth)
    (0, 8), (0, 16), (0, 1), (0, 2), (0, 4), (2, 8), (2, 16), (3, 8), (3, 4), (4, 8), (6, 8), (6, 16),
]


def expected_bgr(samples, ct, bd, palette):
    s = samples.astype(np.int64)
    if ct == 3:
        rgb = np.asarray(palette, np.uint8)[s[..., 0]]
    else:
        v = (s >> 8) if bd == 16 else ((s * 255 // ((1 << bd) - 1)) if (bd < 8 and ct == 0) else s)
        rgb = np.repeat(v[..., :1], 3, axis=2) if ct in (0, 4) else v[..., :3]
    return np.ascontiguousarray(rgb[..., ::-1]).astype(np.uint8)


@pytest.mark.parametrize("ct,bd", CASES)
def test_png_colour_decode_all_types_and_filters(lib, ct, bd):
    rng = np.random.default_rng(100 * ct + bd)
    h, w = 23, 37  # odd sizes: partial bytes for the sub-byte depths
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ct]
    palette = rng.integers(0, 256, (min(256, 1 << bd), 3)) if ct == 3 else None
    hi = len(palette) if ct == 3 else (1 << bd)
    # smooth + noise so that every filter type produces non-trivial residuals
    base = (np.add.outer(np.arange(h), np.arange(w))[..., None] * (hi // 64 + 1) + rng.integers(0, max(2, hi // 8), (h, w, ch))) % hi
    png = encode_png(base, ct, bd, palette=palette, extra_chunks=[(b"gAMA", b"\x00\x00\xb1\x8f"), (b"tEXt", b"Comment\x00synthetic")])
    exp = expected_bgr(base, ct, bd, palette)
    assert np.array_equal(lib.decode_color(png), exp)
    assert np.array_equal(io_oracle.decode_color(png), exp)


def test_png_depth16_decode_and_files(lib, tmp_path):
    rng = np.random.default_rng(5)
    depth = (1500 + 40 * np.add.outer(np.arange(48), np.arange(64)) % 3000 + rng.integers(0, 9, (48, 64))).astype(np.uint16)
    depth[5:9, 7:30] = 0
    depth[0, 0], depth[0, 1] = 65535, 256  # both bytes matter: big-endian on disk, host order in memory
    for filters in ((0,), (1,), (2,), (3,), (4,), (4, 1, 3, 2, 0)):
        png = encode_png(depth, 0, 16, filters=filters, idat_split=5)
        assert np.array_equal(lib.decode_depth16(png), depth)
        assert np.array_equal(io_oracle.decode_depth16(png), depth)
    p = tmp_path / "d.png"
    p.write_bytes(encode_png(depth, 0, 16))
    assert np.array_equal(lib.imread_depth16(str(p)), depth)
    g8 = (depth >> 8).astype(np.uint8)
    assert np.array_equal(lib.decode_depth16(encode_png(g8, 0, 8)), g8.astype(np.uint16))  # 8-bit grey is widened
    rgb = rng.integers(0, 256, (48, 64, 3))
    c = tmp_path / "c.png"
    c.write_bytes(encode_png(rgb, 2, 8))
    assert np.array_equal(lib.imread_color(str(c)), rgb[..., ::-1].astype(np.uint8))  # B G R like cv::imread
    with pytest.raises(sfio.SfIoError):
        lib.decode_depth16(encode_png(rgb, 2, 8))  # a colour image is not a depth image


def test_png_errors(lib, tmp_path):
    good = encode_png(np.arange(64, dtype=np.uint8).reshape(8, 8), 0, 8)
    with pytest.raises(sfio.SfIoError):
        lib.decode_color(b"JFIF" + good[4:])  # signature
    bad = bytearray(good)
    bad[40] ^= 0x55
    with pytest.raises(sfio.SfIoError):
        lib.decode_color(bytes(bad))  # CRC
    with pytest.raises(sfio.SfIoError):
        lib.decode_color(good[:-20])  # truncated
    with pytest.raises(sfio.SfIoError):
        lib.decode_color(encode_png(np.zeros((8, 8), np.uint8), 0, 8, interlace=1))  # Adam7: unsupported, reported
    with pytest.raises(sfio.SfIoError):
        lib.imread_color(str(tmp_path / "missing.png"))


def test_assoc_file(lib, tmp_path):
    d = str(tmp_path) + "/"
    text = ("# color and depth association\n"
            "\n"
            "1305031102.175304 rgb/1305031102.175304.png 1305031102.160407 depth/1305031102.160407.png\n"
            "1305031102.211214 rgb/1305031102.211214.png 1305031102.194330 depth/1305031102.194330.png\n"
            "#1305031102.243211 rgb/skipped.png 1 depth/skipped.png\n"
            "1305031102.275326   rgb/c.png\t1305031102.262886 depth/d.png   trailing tokens are ignored\n"
            "not-a-number rgb/x.png 3 depth/x.png\n"
            "1305031103.0 rgb/after_the_break.png 1305031103.0 depth/after_the_break.png\n")
    (tmp_path / "rgbd_assoc.txt").write_text(text)
    ts, fd, fc = lib.load_assoc(d, "rgbd_assoc.txt")
    assert ts == [1305031102.160407, 1305031102.194330, 1305031102.262886]  # the DEPTH timestamps
    assert fd == [d + "depth/1305031102.160407.png", d + "depth/1305031102.194330.png", d + "depth/d.png"]
    assert fc == [d + "rgb/1305031102.175304.png", d + "rgb/1305031102.211214.png", d + "rgb/c.png"]
    assert (ts, fd, fc) == io_oracle.load_assoc(d, "rgbd_assoc.txt")
    with pytest.raises(sfio.SfIoError):
        lib.load_assoc(d, "missing.txt")


def random_pose(rng, angle_scale=3.0):
    w = rng.normal(0, 1, 3)
    w = w / np.linalg.norm(w) * rng.uniform(0, angle_scale)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + (np.sin(th) / th) * K + ((1 - np.cos(th)) / th ** 2) * K @ K if th > 1e-9 else np.eye(3)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = rng.normal(0, 2, 3)
    return T.astype(np.float32)


def test_trajectory_lines_and_pose_composition(lib):
    rng = np.random.default_rng(11)
    branches = set()
    pose = np.eye(4, dtype=np.float32)
    for k in range(300):
        T = random_pose(rng, 3.1 if k % 3 else 0.05)
        a, b = lib.pose_compose(pose, T), io_oracle.pose_compose(pose, T)
        assert np.array_equal(a, b)
        pose = a if k % 50 else np.eye(4, dtype=np.float32)
        tr = np.trace(pose[:3, :3])
        branches.add(bool(tr > 0))
        ts = 1305031102.0 + 0.033 * k
        for rot in (0, 1):
            assert lib.trajectory_line(ts, pose, rot) == io_oracle.trajectory_line(ts, pose, rot), (k, rot)
    assert branches == {True, False}  # both quaternion branches of Eigen's conversion were exercised
    line = lib.trajectory_line(12.5, np.eye(4, dtype=np.float32), 0)
    assert line == "12.500000 0 0 0 0 0 0 1\n"
    z = lib.trajectory_line(12.5, np.eye(4, dtype=np.float32), 1).split()
    assert z[0] == "12.5000" and abs(float(z[6])) == 1.0 and abs(float(z[7])) < 1e-7  # identity * Rz(pi): q = (0, 0, +-1, ~0)


# ------------------------------------------------------------------------------------------------
#  end to end: a synthetic dataset on disk -> loader -> input stage -> solver -> trajectory file
# ------------------------------------------------------------------------------------------------
def write_dataset(root, n_frames, sphere=True):
    """TUM-style directory of a synthetic walk (the reference README.md:67-89 layout); returns the ground-truth poses."""
    from staticfusion_amd.synth import DEFAULT_XI, Scene, se3_exp

    os.makedirs(os.path.join(root, "rgb"))
    os.makedirs(os.path.join(root, "depth"))
    scene = Scene(seed=77, sphere=sphere)
    xi = np.array(DEFAULT_XI) * 0.6
    T, gts, lines = np.eye(4), [], ["# color depth association, synthetic"]
    for k in range(n_frames):
        depth, inten = scene.render(T, 640, 480, sphere_offset=(0.02 * k, 0, 0))
        d_mm = np.clip(np.rint(depth * 1000.0), 0, 65535).astype(np.uint16)
        g8 = np.clip(np.rint(inten * 255.0), 0, 255).astype(np.uint8)
        t = 1305031102.0 + k / 30.0
        # the loader flips vertically (FrontEnd.cpp:231): store the frames upside down so that it sees them upright
        open(os.path.join(root, "rgb", "%.6f.png" % t), "wb").write(encode_png(np.repeat(g8[::-1, :, None], 3, axis=2), 2, 8))
        open(os.path.join(root, "depth", "%.6f.png" % t), "wb").write(encode_png(d_mm[::-1], 0, 16))
        lines.append("%.6f rgb/%.6f.png %.6f depth/%.6f.png" % (t, t, t, t))
        gts.append(T.copy())
        T = T @ se3_exp(xi)
    open(os.path.join(root, "rgbd_assoc.txt"), "w").write("\n".join(lines) + "\n")
    return gts


def test_sequence_runner_on_the_oracle_tiny(ora, lib, tmp_path):
    """CPU-sized: 3 frames through loadAssoc -> PNG -> loader -> bilateral filter -> solver -> trajectory lines."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from run_sequence import run
    from staticfusion_amd.synth import pose_delta

    root = str(tmp_path / "ds")
    gts = write_dataset(root, 3)
    poses, lines, _ = run(ora, lib, root, out_path=str(tmp_path / "traj.txt"))
    assert len(poses) == 3 and open(str(tmp_path / "traj.txt")).read() == "".join(lines)
    for k in range(1, 3):  # tracks the synthetic camera (frame-to-frame, quantised, filtered input): centimetre / 0.1 deg level
        rot, trans = pose_delta(poses[k], gts[k])
        assert rot < 5e-3 and trans < 1.5e-2, (k, rot, trans)
    assert lines[0].split()[1:] == ["0", "0", "0", "0", "0", "0", "1"]


@pytest.mark.gpu
def test_sequence_runner_hip_vs_oracle(hip, ora, lib, tmp_path):
    """8 frames from disk on MI355X: the per-frame pose is the CPU path's (<= 1e-4), the trajectory files agree."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from run_sequence import run
    from staticfusion_amd.synth import pose_delta

    root = str(tmp_path / "ds")
    gts = write_dataset(root, 8)
    pg, lg, sg = run(hip, lib, root)
    po, lo, so = run(ora, lib, root)
    for k in range(8):
        rot, trans = pose_delta(po[k], pg[k])
        assert rot <= 1e-4 and trans <= 1e-4, (k, rot, trans)
    assert np.array_equal(sg.labels(0), so.labels(0))  # cluster labels of the last frame: bit exact
    for a, b in zip(lg, lo):  # 6 significant digits printed
        assert np.allclose([float(x) for x in a.split()], [float(x) for x in b.split()], rtol=0, atol=2e-5)
    rot, trans = pose_delta(pg[7], gts[7])
    assert rot < 2e-2 and trans < 5e-2


def test_io_header_matches_library_exports():
    """every function include/sf_io.h declares is exported by libsf_io.so and bound by staticfusion_amd/io.py -- and nothing else"""
    import re
    import subprocess

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    hdr = open(os.path.join(root, "include", "sf_io.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(sf_io_\w+)\s*\(", hdr)))
    assert declared == sorted(sfio.SIGNATURES.keys())
    out = subprocess.check_output(["nm", "-D", "--defined-only", sfio.LIB]).decode()
    assert set(re.findall(r" T (sf_io_\w+)", out)) == set(declared)


def build_example(tmp_path):
    import subprocess

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    libdir = os.path.join(root, "staticfusion_amd", "csrc")
    exe = str(tmp_path / "imagesequence_driver")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "imagesequence_driver.cpp"),
                           "-o", exe, "-L" + libdir, "-lsf_hip", "-lsf_io", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_example_driver_builds(tmp_path):
    assert os.path.exists(build_example(tmp_path))


@pytest.mark.gpu
def test_cpp_example_driver_matches_the_python_runner(hip, lib, tmp_path):
    """examples/imagesequence_driver.cpp (the reference's image-sequence main loop over StaticFusionCompat + sf_io)
    writes the same trajectory file as tools/run_sequence.py: both are the same calls on the same device code."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from run_sequence import run

    root = str(tmp_path / "ds")
    write_dataset(root, 7)
    exe = build_example(tmp_path)
    out = str(tmp_path / "cpp.freiburg")
    # the example calls sf_create (SF_VARIANT_AUTO), which honours SF_VARIANT: same build of the frame kernel on both sides
    subprocess.check_call([exe, root, out], env=dict(os.environ, SF_VARIANT=hip.default_variant))
    _, lines, _ = run(hip, lib, root)
    assert open(out).read() == "".join(lines)


@pytest.mark.gpu
def test_sequence_runner_keyframe_model_mode(hip, ora, lib, tmp_path):
    """frame-to-MODEL tracking from disk: the surfel model of frame 0 (GlobalModel::initialise) is rendered at the
    current pose for every prediction (getPredictedImages). HIP == oracle per frame; the pose error against the
    synthetic ground truth does not accumulate as it does frame to frame."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from run_sequence import run
    from staticfusion_amd.synth import pose_delta

    root = str(tmp_path / "ds")
    gts = write_dataset(root, 10, sphere=False)
    po, lo, _ = run(ora, lib, root, mode="keyframe")
    # The depth test of the rendering flips single pixels for a 1e-7 change of the pose (true of the reference's GL
    # rasteriser too), and the pose feeds back into the next prediction: the per-frame parity is therefore checked
    # with both implementations rendering at the SAME (the oracle's) poses ...
    pt, _, _ = run(hip, lib, root, mode="keyframe", prediction_poses=po)
    worst = (0.0, 0.0)
    for k in range(1, 10):  # per-frame motion = the solver's output
        rel_o = np.linalg.inv(po[k - 1].astype(np.float64)) @ po[k]
        rel_g = np.linalg.inv(pt[k - 1].astype(np.float64)) @ pt[k]
        rot, trans = pose_delta(rel_o, rel_g)
        worst = (max(worst[0], rot), max(worst[1], trans))
    assert worst[0] <= 1e-4 and worst[1] <= 1e-4, worst
    # ... and the free-running trajectories stay together at the millimetre / 0.05 degree level
    pg, lg, _ = run(hip, lib, root, mode="keyframe")
    free = pose_delta(po[9], pg[9])
    print("worst per-frame HIP vs oracle on identical predictions", worst, "; free-running after 9 frames", free)
    assert free[0] < 1e-3 and free[1] < 2e-3, free
    pf, _, _ = run(hip, lib, root, mode="frame")
    err_model = pose_delta(pg[9], gts[9])
    err_frame = pose_delta(pf[9], gts[9])
    assert err_model[0] < 1e-2 and err_model[1] < 3e-2, err_model
    print("pose error after 9 frames: keyframe model", err_model, "frame-to-frame", err_frame)


def test_sequence_runner_fusion_mode_on_the_oracle(ora, lib, tmp_path):
    """CPU-sized: the reference's full loop (solve -> fuseFrame -> getPredictedImages) on 5 synthetic frames."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from run_sequence import run
    from staticfusion_amd.synth import pose_delta

    root = str(tmp_path / "ds")
    gts = write_dataset(root, 5, sphere=False)
    trace = []
    poses, lines, s = run(ora, lib, root, mode="fusion", trace=trace)
    assert len(poses) == 5 and len(lines) == 5 and [t["frame"] for t in trace] == [1, 2, 3, 4]
    assert trace[0]["stats"][0] == 0 and trace[0]["count"] > 70000           # the bootstrap fuse initialises the map
    assert all(t["stats"][1] > 0.95 * t["stats"][0] > 0 for t in trace[1:])   # then nearly every candidate pixel finds its surfel
    info = s.map.info()
    assert info["tick"] == 5 and np.array_equal(info["pose"], poses[-1])
    acc = np.eye(4, dtype=np.float32)
    for T in s.increments[1:]:
        acc = lib.pose_compose(acc, T)
    assert np.array_equal(acc, poses[-1])                                     # currPose = product of the T_odometry
    err = pose_delta(poses[-1], gts[-1])
    assert err[0] < 5e-3 and err[1] < 2e-2, err


@pytest.mark.gpu
def test_sequence_runner_fusion_mode_hip_vs_oracle(hip, ora, lib, tmp_path):
    """the full loop without OpenGL on the GPU. The per-frame parity is checked with the HIP run fusing at the oracle's
    T_odometry (its map and therefore its predictions then see the same poses); free-running the two trajectories stay
    together, and the fused map tracks better than frame-to-frame odometry."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from run_sequence import run
    from staticfusion_amd.synth import pose_delta

    root = str(tmp_path / "ds")
    gts = write_dataset(root, 10, sphere=True)
    tro, trt = [], []
    po, _, so = run(ora, lib, root, mode="fusion", trace=tro)
    pt, _, st = run(hip, lib, root, mode="fusion", odometry_override=so.increments, trace=trt)
    worst = (0.0, 0.0)
    for a, b in zip(tro, trt):
        rot, trans = pose_delta(a["T"], b["T"])
        worst = (max(worst[0], rot), max(worst[1], trans))
        assert abs(a["count"] - b["count"]) <= 0.002 * a["count"] and abs(a["stats"][1] - b["stats"][1]) <= 0.01 * a["stats"][0], (a, b)
    print("worst per-frame T_odometry difference HIP vs oracle, teacher-forced fusion:", worst)
    assert worst[0] <= 1e-4 and worst[1] <= 1e-4, worst
    assert all(np.array_equal(x, y) for x, y in zip(po, pt))  # same increments in, same poses out
    pg, _, sg = run(hip, lib, root, mode="fusion")
    free = pose_delta(po[-1], pg[-1])
    err_map, err_gt_o = pose_delta(pg[-1], gts[-1]), pose_delta(po[-1], gts[-1])
    pf, _, _ = run(hip, lib, root, mode="frame")
    err_frame = pose_delta(pf[-1], gts[-1])
    print("free-running HIP vs oracle after 9 frames", free, "; error vs ground truth: fused map", err_map, "oracle", err_gt_o, "frame-to-frame", err_frame)
    assert free[0] < 1e-3 and free[1] < 2e-3, free
    assert err_map[0] < 1e-2 and err_map[1] < 3e-2, err_map


def build_headless(tmp_path):
    import subprocess

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    libdir = os.path.join(root, "staticfusion_amd", "csrc")
    exe = str(tmp_path / "staticfusion_headless")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "staticfusion_headless.cpp"),
                           "-o", exe, "-L" + libdir, "-lsf_hip", "-lsf_io", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_headless_example_builds(tmp_path):
    assert os.path.exists(build_headless(tmp_path))


def test_ply_writer_matches_the_restatement(lib, tmp_path):
    """Reconstruction::savePly (Reconstruction.cpp:358-455): header text, confidence gate, colour decode, negated normals"""
    rng = np.random.default_rng(3)
    s = rng.normal(size=(2000, 12)).astype(np.float32)
    s[:, 3] = rng.uniform(0, 1, 2000)
    s[:, 4] = rng.integers(0, 1 << 24, 2000)
    s[7, 3] = np.float32(0.25)  # not above the threshold
    path = str(tmp_path / "map.ply")
    n = lib.save_ply(path, s, 0.25)
    data = open(path, "rb").read()
    assert n == int((s[:, 3] > np.float32(0.25)).sum()) and data == io_oracle.save_ply_bytes(s, 0.25)
    assert data.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\n" % n)
    assert len(data) == data.index(b"end_header\n") + 11 + n * 31
    assert lib.save_ply(str(tmp_path / "empty.ply"), np.zeros((0, 12), np.float32), 0.25) == 0


@pytest.mark.gpu
def test_headless_example_is_the_python_fusion_loop(hip, lib, tmp_path):
    """examples/staticfusion_headless.cpp (the reference's full image-sequence loop over StaticFusionCompat +
    ReconstructionCompat) writes the trajectory of tools/run_sequence.py --mode fusion and the PLY of its final map."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from run_sequence import run

    root = str(tmp_path / "ds")
    write_dataset(root, 8)
    exe = build_headless(tmp_path)
    prefix = str(tmp_path / "out")
    msg = subprocess.check_output([exe, root, prefix], env=dict(os.environ, SF_VARIANT=hip.default_variant)).decode()
    _, lines, s = run(hip, lib, root, mode="fusion")
    assert open(prefix + ".freiburg").read() == "".join(lines), msg
    assert open(prefix + ".ply", "rb").read() == io_oracle.save_ply_bytes(s.map.download(), 0.25), msg
    assert "8 frames" in msg
