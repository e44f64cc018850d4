"""Synthetic VGA -> QVGA RGB-D pairs (BASELINE.json configs[1] and configs[2]).

Scene: a room of analytic planes (back wall z = 3.0 m tilted 10 deg about y, floor y = +1.0 m,
side wall x = -1.6 m), optionally a sphere (r = 0.25 m at (0.2, 0.1, 1.5) m) that moves in the
world between the two frames.  Texture is a band-limited sum of six plane waves in WORLD
coordinates (wave vectors 2..12 rad/m, phases from a self-implemented 64-bit LCG), so both views
see the same surface pattern.  Both views are ray-cast exactly at 640x480 with the pinhole model
the solver assumes (one focal length f = W / (2 tan(fovh/2)) for both axes, reference
FrontEnd.cpp:378-386), depth is quantised to uint16 millimetres and intensity to 8 bit like the
reference's loaders (FrontEnd.cpp:231-243), then decimated [::2, ::2] (FrontEnd.cpp:228-246).

Camera convention (reference FrontEnd.cpp:800-816): T_odometry maps points of the NEW camera
frame into the OLD (prediction) frame, i.e. it is the pose of the new camera in the old frame.
`make_pair` returns (old image, new image, T_gt) with T_gt = exp(xi^).
"""
import numpy as np

FOVH = float(np.float32(np.pi * 62.5 / 180.0))
DEFAULT_XI = (0.010, -0.005, 0.008, 0.004, -0.006, 0.003)


class LCG64:
    """Knuth MMIX linear congruential generator; uniform() in [0, 1)."""

    MASK = (1 << 64) - 1

    def __init__(self, seed):
        self.x = seed & self.MASK

    def next_u64(self):
        self.x = (6364136223846793005 * self.x + 1442695040888963407) & self.MASK
        return self.x

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * ((self.next_u64() >> 11) / float(1 << 53))


def se3_exp(xi):
    """4x4 rigid transform of the twist (vx, vy, vz, wx, wy, wz), float64."""
    xi = np.asarray(xi, dtype=np.float64)
    v, w = xi[:3], xi[3:]
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-9:
        R = np.eye(3) + K
        V = np.eye(3) + 0.5 * K
    else:
        a = np.sin(th) / th
        b = (1 - np.cos(th)) / th**2
        c = (th - np.sin(th)) / th**3
        R = np.eye(3) + a * K + b * K @ K
        V = np.eye(3) + b * K + c * K @ K
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ v
    return T


def rotation_angle(R):
    c = np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0)
    s = 0.5 * np.sqrt((R[2, 1] - R[1, 2]) ** 2 + (R[0, 2] - R[2, 0]) ** 2 + (R[1, 0] - R[0, 1]) ** 2)
    return float(np.arctan2(s, c))


def pose_delta(Ta, Tb):
    """(rotation angle [rad], translation norm [m]) of Ta^-1 Tb — the metric of BASELINE.json."""
    D = np.linalg.inv(np.asarray(Ta, dtype=np.float64)) @ np.asarray(Tb, dtype=np.float64)
    return rotation_angle(D[:3, :3]), float(np.linalg.norm(D[:3, 3]))


class Texture:
    def __init__(self, seed, n_waves=6):
        g = LCG64(seed)
        ks, ph = [], []
        for _ in range(n_waves):
            d = np.array([g.uniform(-1, 1), g.uniform(-1, 1), g.uniform(-1, 1)])
            d /= max(np.linalg.norm(d), 1e-6)
            ks.append(d * g.uniform(2.0, 12.0))
            ph.append(g.uniform(0.0, 2 * np.pi))
        self.k = np.array(ks)
        self.phi = np.array(ph)

    def __call__(self, p):  # p: (..., 3)
        s = np.sin(p @ self.k.T + self.phi).sum(axis=-1)
        return np.clip(0.5 + 0.15 * s, 0.05, 0.95)


class Scene:
    def __init__(self, seed=1234, sphere=False, sphere_seed=5678, wall_z=3.0, tilt_deg=10.0,
                 floor_y=1.0, side_x=-1.6):
        t = np.deg2rad(tilt_deg)
        n_back = np.array([np.sin(t), 0.0, np.cos(t)])
        self.planes = [
            (n_back, float(n_back @ np.array([0.0, 0.0, wall_z]))),
            (np.array([0.0, 1.0, 0.0]), floor_y),
            (np.array([1.0, 0.0, 0.0]), side_x),
        ]
        self.tex = Texture(seed)
        self.sphere = sphere
        self.sphere_c0 = np.array([0.2, 0.1, 1.5])
        self.sphere_r = 0.25
        self.sphere_tex = Texture(sphere_seed)

    def render(self, T_cam, W=640, H=480, sphere_offset=(0.0, 0.0, 0.0), fovh=FOVH, stride=1):
        """Ray-cast from a camera with pose T_cam (camera -> world). Returns (depth[m], intensity) HxW float64.
        stride = 2 casts only the rays of the pixels a [::2, ::2] decimation keeps (H/2 x W/2 output)."""
        f = W / (2.0 * np.tan(0.5 * fovh))
        cx, cy = W / 2.0 - 1.0, H / 2.0 - 1.0  # decimation-consistent principal point (see module doc)
        u, v = np.meshgrid(np.arange(0, W, stride, dtype=np.float64), np.arange(0, H, stride, dtype=np.float64))
        rays_c = np.stack([(u - cx) / f, (v - cy) / f, np.ones_like(u)], axis=-1)
        R, o = T_cam[:3, :3], T_cam[:3, 3]
        d = rays_c @ R.T
        t_best = np.full(u.shape, np.inf)
        which = np.full(u.shape, -1, dtype=np.int32)
        for idx, (n, d0) in enumerate(self.planes):
            den = d @ n
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (d0 - o @ n) / den
            ok = (np.abs(den) > 1e-12) & (t > 1e-6) & (t < t_best)
            t_best = np.where(ok, t, t_best)
            which = np.where(ok, idx, which)
        inten = None
        if self.sphere:
            c = self.sphere_c0 + np.asarray(sphere_offset, dtype=np.float64)
            oc = o - c
            a = (d * d).sum(-1)
            b = 2.0 * (d @ oc)
            cc = oc @ oc - self.sphere_r**2
            disc = b * b - 4 * a * cc
            with np.errstate(invalid="ignore"):
                ts = (-b - np.sqrt(disc)) / (2 * a)
            ok = (disc > 0) & (ts > 1e-6) & (ts < t_best)
            t_best = np.where(ok, ts, t_best)
            which = np.where(ok, 99, which)
        hit = np.isfinite(t_best)
        t_safe = np.where(hit, t_best, 0.0)
        p = o + d * t_safe[..., None]
        inten = self.tex(p)
        if self.sphere:
            c = self.sphere_c0 + np.asarray(sphere_offset, dtype=np.float64)
            inten = np.where(which == 99, self.sphere_tex((p - c) * 4.0), inten)
        depth = np.where(hit, t_safe, 0.0)  # rays_c has z = 1, so the ray parameter IS the camera-frame depth
        inten = np.where(hit, inten, 0.0)
        return depth, inten


def quantise_and_decimate(depth, inten, max_depth=None, decimate=True):
    """uint16 mm depth / 8-bit grey, like the reference's loaders, then [::2, ::2] (decimate=False: the images were
    rendered with stride 2 -- only the rays that survive the decimation, the same values)."""
    d_mm = np.clip(np.rint(depth * 1000.0), 0, 65535).astype(np.uint16)
    if max_depth is not None:
        d_mm = np.where(depth < max_depth, d_mm, 0).astype(np.uint16)
    d = (d_mm.astype(np.float64) * 0.001).astype(np.float32)  # convertTo(CV_32FC1, 1.0/1000.0), FrontEnd.cpp:243
    g8 = np.clip(np.rint(inten * 255.0), 0, 255).astype(np.uint8)
    x = g8.astype(np.float32) * np.float32(1.0 / 255.0)
    i = np.float32(0.299) * x + np.float32(0.587) * x + np.float32(0.114) * x  # FrontEnd.cpp:236
    if not decimate:
        return np.ascontiguousarray(d), np.ascontiguousarray(i.astype(np.float32))
    return np.ascontiguousarray(d[::2, ::2]), np.ascontiguousarray(i[::2, ::2].astype(np.float32))


def make_pair(seed=1234, xi=DEFAULT_XI, sphere=False, sphere_motion=(0.05, 0.0, 0.0), out_rows=240, out_cols=320):
    """One RGB-D pair. Returns dict(old=(depth, intensity), new=(depth, intensity), T_gt, xi).

    `old` plays the role of the prediction image (reference depthPrediction), `new` the current one.
    Images are (rows, cols) float32.  out_rows/out_cols other than 240x320 render at 2x that size.
    """
    W, H = 2 * out_cols, 2 * out_rows
    scene = Scene(seed=seed, sphere=sphere, sphere_seed=seed + 4444)
    T0 = np.eye(4)
    T1 = se3_exp(xi)
    d0, i0 = scene.render(T0, W, H)
    d1, i1 = scene.render(T1, W, H, sphere_offset=sphere_motion if sphere else (0, 0, 0))
    return {
        "old": quantise_and_decimate(d0, i0),
        "new": quantise_and_decimate(d1, i1),
        "T_gt": T1,
        "xi": np.asarray(xi, dtype=np.float64),
    }


def make_batch(n, base_seed=1234, sphere=False, distinct=None, out_rows=240, out_cols=320):
    """n pairs; `distinct` (default min(n, 16)) different scenes/motions are generated and tiled."""
    distinct = min(n, 16) if distinct is None else min(n, distinct)
    pairs = []
    for j in range(distinct):
        g = LCG64(base_seed + 7919 * j)
        scale = g.uniform(0.5, 1.5)
        xi = np.array(DEFAULT_XI) * scale * np.array([g.uniform(0.6, 1.4) * (1 if g.uniform() < 0.5 else -1) for _ in range(6)])
        pairs.append(make_pair(seed=base_seed + j, xi=xi, sphere=sphere, out_rows=out_rows, out_cols=out_cols))
    return [pairs[i % distinct] for i in range(n)]


def sequence_trajectory(seed, frames, scale=1.0):
    """Camera poses (camera -> world, 4x4 float64) of a smooth random walk: the per-frame twist is a first-order
    autoregressive process around DEFAULT_XI-sized steps (about 1 cm and 0.3 degrees per frame), so consecutive frames
    overlap like a hand-held camera's, and no two sequences (seeds) move alike. Returns (poses, sphere offsets)."""
    g = LCG64(0x5EED0000 + seed)
    sigma = np.array(DEFAULT_XI) * scale * 1.2
    xi = np.array([g.uniform(-1, 1) for _ in range(6)]) * sigma
    T = np.eye(4)
    poses, offsets = [], []
    ph = [g.uniform(0, 2 * np.pi) for _ in range(3)]
    for k in range(frames):
        poses.append(T.copy())
        # the sphere swings through the scene (a few cm per frame at most)
        offsets.append((0.30 * np.sin(0.07 * k + ph[0]), 0.10 * np.sin(0.05 * k + ph[1]), 0.15 * np.sin(0.04 * k + ph[2])))
        noise = np.array([g.uniform(-1, 1) for _ in range(6)]) * sigma
        xi = 0.9 * xi + 0.45 * noise
        # keep the camera in the room: a weak pull of the position back to the origin
        pull = -0.02 * np.concatenate([T[:3, :3].T @ T[:3, 3], np.zeros(3)])
        T = T @ se3_exp(xi + pull)
    return poses, offsets


def _render_sequence_frame(args):
    seed, pose, offset, sphere, W, H = args
    scene = Scene(seed=seed, sphere=sphere, sphere_seed=seed + 4444)
    # stride 2: only the rays of the pixels the loaders' decimation keeps (every operation of render() is per pixel: the same
    # bits as rendering all of them and dropping three quarters, tests/test_capi_and_host.py; 0.03 s instead of 0.12 s a frame)
    return quantise_and_decimate(*scene.render(pose, W, H, sphere_offset=offset if sphere else (0, 0, 0), stride=2), decimate=False)


def make_sequence(seed, frames, sphere=True, out_rows=240, out_cols=320, scale=1.0, pool=None):
    """One synthetic RGB-D sequence: `frames` (depth, intensity) QVGA images rendered at 2x the output size along
    sequence_trajectory(seed). T_gt[k] = pose_{k-1}^-1 pose_k is what the solver should report for frame k against frame
    k - 1 as prediction. `pool`: an optional multiprocessing pool for the ray casting (0.03 s per frame and core)."""
    poses, offsets = sequence_trajectory(seed, frames, scale)
    jobs = [(seed, poses[k], offsets[k], sphere, 2 * out_cols, 2 * out_rows) for k in range(frames)]
    imgs = pool.map(_render_sequence_frame, jobs, chunksize=4) if pool is not None else [_render_sequence_frame(j) for j in jobs]
    T_gt = [np.eye(4)] + [np.linalg.inv(poses[k - 1]) @ poses[k] for k in range(1, frames)]
    return {"frames": imgs, "poses": poses, "T_gt": T_gt}


def _source_tag():
    """A short hash of this file: cached renderings belong to the generator that made them."""
    import hashlib

    with open(__file__.replace(".pyc", ".py"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:10]


def sequence_arrays(seed, frames, out_rows=240, out_cols=320, pool=None, cache_dir=None):
    """make_sequence(seed, frames) in the layout of the solver's HBM frame pools: (depth [frames][n0], intensity [frames][n0],
    T_gt [frames][4][4]) with column-major images. With `cache_dir` a sequence rendered before is read back from there; the
    file name carries the seed, the frame count, the resolution and a hash of this generator, the shape is checked on load
    and the file appears by an atomic rename (several ranks of a node may render the same seed at once: last rename wins,
    every reader sees a complete file)."""
    import os

    n0 = out_rows * out_cols
    path = None
    if cache_dir:
        path = os.path.join(cache_dir, "sf_seq_%s_%d_%d_%dx%d_u%d.npz" % (_source_tag(), seed, frames, out_rows, out_cols, os.getuid()))
        try:
            with np.load(path) as z:
                if z["d"].shape == (frames, n0) and z["i"].shape == (frames, n0):
                    return z["d"], z["i"], z["T_gt"]
        except Exception:
            pass
    seq = make_sequence(seed, frames, sphere=True, out_rows=out_rows, out_cols=out_cols, pool=pool)
    col = lambda a: np.ascontiguousarray(np.asarray(a, np.float32).T).ravel()
    d, i = np.stack([col(f[0]) for f in seq["frames"]]), np.stack([col(f[1]) for f in seq["frames"]])
    T_gt = np.stack(seq["T_gt"])
    if path:
        try:
            tmp = "%s.%d.tmp.npz" % (path, os.getpid())
            np.savez(tmp, d=d, i=i, T_gt=T_gt)
            os.replace(tmp, path)
        except Exception:
            pass
    return d, i, T_gt
