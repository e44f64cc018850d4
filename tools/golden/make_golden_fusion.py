"""Independent Python (float32 scalars) derivation of the FUSION half of Reconstruction::fuseFrame, used
to emit tests/golden/fusion_40x30.npz.

Written from the reference's shaders, not from the C++ oracle or the HIP kernels, and sharing no code with them: one GLSL
statement at a time, texture fetches through float texture coordinates (where the oracle short-cuts to integer neighbours),
the update maps as a dictionary keyed by texel, transform feedback as Python lists.

  predict_indices   IndexMap::predictIndices       reference IndexMap.cpp:117-184, Shaders/index_map.vert, index_map.frag
  data_pass         GlobalModel::fuse, first half  GlobalModel.cpp:322-425, Shaders/data.vert, data.geom, data.frag
  update_pass       GlobalModel::fuse, second half GlobalModel.cpp:427-492, Shaders/update.vert
  clean_pass        GlobalModel::clean             GlobalModel.cpp:494-601, Shaders/copy_unstable.vert, copy_unstable.geom
  weighting         Reconstruction::fuseFrame      Reconstruction.cpp:264-282 (rotation vector by scipy instead of rodrigues2)
  init_model        GlobalModel::initialise        GlobalModel.cpp:200-258, Reconstruction::computeFeedbackBuffers (:205-216),
                                                   Shaders/FeedbackBuffer.cpp, vertex_feedback.vert/.geom, init_unstable.vert
  predict_images    Reconstruction::getPredictedImages  Reconstruction.cpp:628-720, IndexMap.cpp:221-300, Shaders/splat.vert,
                                                   combo_splat.frag, fill_vertex.frag, fill_vertex_from_texture.frag, fill_rgb.frag,
                                                   extract_depth.frag, Shaders/Resize.cpp + denseEnough (:218-233)

exp / log are NumPy's (the oracle uses include/sf_detmath.h's fmaf sequences), the pose inverse is numpy.linalg.inv: the
fixture is compared with a small tolerance on the floats and EXACTLY on everything discrete (counts, which pixel merged into
which surfel, the index image, which surfels survive the cleaning).

The frames go through the independent input-stage derivation (make_golden_input.py) and buildSegmImage (make_golden.py).

Run (in the build container):  python tools/golden/make_golden_fusion.py  -> tests/golden/fusion_40x30.npz
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from make_golden import segm_image  # noqa: E402
from make_golden_input import bilateral_mm, load_frame, metricise  # noqa: E402

F = np.float32
ROWS, COLS, RES = 30, 40, 2
FACTOR = 4           # IndexMap::FACTOR
TEX_DIM = 3072       # GlobalModel::TEXTURE_DIMENSION


def v3(a, b, c):
    return np.array([a, b, c], F)


def length(v):
    return F(math.sqrt(F(F(F(v[0] * v[0]) + F(v[1] * v[1])) + F(v[2] * v[2]))))


def normalize(v):
    n = length(v)
    with np.errstate(all="ignore"):
        return np.array([v[0] / n, v[1] / n, v[2] / n], F)


def cross(a, b):
    return v3(F(a[1] * b[2]) - F(b[1] * a[2]), F(a[2] * b[0]) - F(b[2] * a[0]), F(a[0] * b[1]) - F(b[0] * a[1]))


def dot(a, b):
    return F(F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2]))


def mat4_point(T, p):  # (T * vec4(p, 1)).xyz
    return v3(*[F(F(F(F(T[r, 0] * p[0]) + F(T[r, 1] * p[1])) + F(T[r, 2] * p[2])) + T[r, 3]) for r in range(3)])


def mat3_vec(T, p):    # mat3(T) * p
    return v3(*[F(F(F(T[r, 0] * p[0]) + F(T[r, 1] * p[1])) + F(T[r, 2] * p[2])) for r in range(3)])


def fetch(img, u, v):
    """textureLod on a GL_NEAREST, clamp-to-edge texture whose rows are the rows of img"""
    h, w = img.shape[:2]
    xi = min(max(int(math.floor(F(F(u) * F(w)))), 0), w - 1)
    yi = min(max(int(math.floor(F(F(v) * F(h)))), 0), h - 1)
    return img[yi, xi]


def encode_color(c):  # color.glsl:19-25
    rgb = int(round(float(F(c[0] * F(255)))))
    rgb = (rgb << 8) + int(round(float(F(c[1] * F(255)))))
    rgb = (rgb << 8) + int(round(float(F(c[2] * F(255)))))
    return F(rgb)


def decode_color(c):  # color.glsl:27-34
    k = int(c)
    return v3(F(F((k >> 16) & 0xFF) / F(255)), F(F((k >> 8) & 0xFF) / F(255)), F(F(k & 0xFF) / F(255)))


# ------------------------------------------------------------------------------------------------
def predict_indices(surf, t_inv, cam, time, time_delta, max_depth):
    W4, H4 = COLS * FACTOR, ROWS * FACTOR
    index = np.zeros((H4, W4), np.uint32)
    zbuf = np.ones((H4, W4), F)
    cx, cy, fx, fy = [F(c * F(FACTOR)) for c in cam]                     # IndexMap.cpp:136-139
    fcols, frows = F(COLS * FACTOR), F(ROWS * FACTOR)
    for s in range(surf.shape[0]):
        h = mat4_point(t_inv, surf[s, 0:3])                              # index_map.vert:38
        if h[2] > max_depth or h[2] < 0 or F(F(time) - surf[s, 7]) > F(time_delta):  # :43
            continue
        with np.errstate(all="ignore"):
            x = F(F(F(F(F(fx * h[0]) / h[2]) + cx) - F(fcols * F(0.5))) / F(fcols * F(0.5)))   # :51
            y = F(F(F(F(F(fy * h[1]) / h[2]) + cy) - F(frows * F(0.5))) / F(frows * F(0.5)))
        z = F(h[2] / max_depth)                                          # :57
        if not (-1 <= x <= 1 and -1 <= y <= 1 and -1 <= z <= 1):          # clipped (a point is clipped by its centre)
            continue
        xw, yw = F(F(x + F(1)) * F(fcols * F(0.5))), F(F(y + F(1)) * F(frows * F(0.5)))   # viewport
        px, py = int(math.floor(xw)), int(math.floor(yw))                # a size-1 point: the pixel that contains it
        if not (0 <= px < W4 and 0 <= py < H4):
            continue
        depth = F(F(z * F(0.5)) + F(0.5))                                # window depth
        if depth < zbuf[py, px]:                                         # GL_LESS, cleared to 1
            zbuf[py, px] = depth
            index[py, px] = s
    return index


def index_textures(surf, t_inv, idx):
    """what index_map.vert/.frag wrote next to the index: vPosHome + conf, colour/time, normalised normal + radius"""
    q = surf[idx]
    return mat4_point(t_inv, q[0:3]), q[3], q[4:8], normalize(mat3_vec(t_inv, q[8:11])), q[11]


def data_pass(fr, pose, t_inv, cam, time, weighting, max_depth, surf, index):
    """data.vert + data.geom over the uv buffer (x outer, y inner); returns the emitted vertices and the update map"""
    cx, cy, fx, fy = cam
    camz, camw = F(1.0 / float(fx)), F(1.0 / float(fy))                   # GlobalModel.cpp:365-368
    fc, fr_ = F(COLS), F(ROWS)
    emitted, update_map = [], {}
    for i in range(COLS):
        for j in range(ROWS):
            tx = F(float(F(i) / fc) + 1.0 / (2 * float(fc)))              # GlobalModel.cpp:81-82
            ty = F(float(F(j) / fr_) + 1.0 / (2 * float(fr_)))
            x, y = F(tx * fc), F(ty * fr_)                                # data.vert:79-80

            def vertex(depth_img, u, v, xx, yy):                          # geometry.glsl:21-25
                z = F(fetch(depth_img, u, v))
                return v3(F(F(F(xx - cx) * z) * camz), F(F(F(yy - cy) * z) * camw), z)

            one_c, one_r = F(F(1.0) / fc), F(F(1.0) / fr_)
            vpl = vertex(fr["raw"], tx, ty, x, y)                         # :83
            world = mat4_point(pose, vpl)                                 # :84
            vf = vertex(fr["fil"], tx, ty, x, y)                          # :87
            prob = F(fetch(fr["b"], tx, ty))                              # :89
            col = fetch(fr["rgb"], tx, ty).astype(F) / F(255)             # :92 (normalised texture)
            color = encode_color(col)
            # :97 getNormal (geometry.glsl:28-40)
            xf = vertex(fr["fil"], F(tx + one_c), ty, F(x + F(1)), y)
            xb = vertex(fr["fil"], F(tx - one_c), ty, F(x - F(1)), y)
            yf = vertex(fr["fil"], tx, F(ty + one_r), x, F(y + F(1)))
            yb = vertex(fr["fil"], tx, F(ty - one_r), x, F(y - F(1)))
            half = lambda a, b: np.array([F(F(a[k] + b[k]) / F(2)) for k in range(3)], F)
            del_x = half(xb, vf) - half(xf, vf)
            del_y = half(yb, vf) - half(yf, vf)
            n_local = normalize(cross(del_x.astype(F), del_y.astype(F)))
            # :98 getRadius (surfels.glsl:19-35)
            mean_focal = F(F(F(F(1.0) / abs(camz)) + F(F(1.0) / abs(camw))) / F(2.0))
            radius = F(F(vf[2] / mean_focal) * F(1.41421356237))
            with np.errstate(all="ignore"):
                radius_n = min(F(F(2.0) * radius), F(radius / abs(n_local[2])))
            n_world = mat3_vec(pose, n_local)
            # :101 confidence (surfels.glsl:37-47)
            dx, dy = F(x - cx), F(y - cy)
            radial = F(F(math.sqrt(F(F(dx * dx) + F(dy * dy)))) / F(200))
            radial_conf = F(math.exp(-float(F(F(radial * radial) / F(F(2) * F(0.72))))))
            conf = min(prob, min(F(weighting), radial_conf))
            t_last, update_id, best = F(0), 0, 0
            neighbours = all(F(fetch(fr["raw"], u, v)) != 0 for u, v in
                             ((F(tx - one_c), ty), (tx, F(ty - one_r)), (F(tx + one_c), ty), (tx, F(ty + one_r))))   # :52-71
            if int(x) % 2 == int(F(time)) % 2 and int(y) % 2 == int(F(time)) % 2 and neighbours and vpl[2] > 0 and vpl[2] <= max_depth:
                counter = 0
                scale = F(FACTOR)
                sx, sy = F(F(F(1.0) / F(fc * scale)) * F(0.5)), F(F(F(1.0) / F(fr_ * scale)) * F(0.5))   # :121-122
                best_dist, wm = F(1000), F(2)
                xl, yl = F(F(x - cx) * camz), F(F(y - cy) * camw)
                lam = F(math.sqrt(F(F(F(xl * xl) + F(yl * yl)) + F(1))))
                ray = v3(xl, yl, F(1))
                u = F(tx - F(F(scale * sx) * wm))
                while u < F(tx + F(F(scale * sx) * wm)):                   # :133
                    v = F(ty - F(F(scale * sy) * wm))
                    while v < F(ty + F(F(scale * sy) * wm)):               # :135
                        cur = int(fetch(index, u, v))                      # :138
                        if cur > 0:
                            pos, _, _, nrm, _ = index_textures(surf, t_inv, cur)
                            if abs(F(F(pos[2] * lam) - F(vpl[2] * lam))) < F(0.05):        # :144
                                dist = F(length(cross(ray, pos)) / length(ray))           # :146
                                ok = abs(nrm[2]) < F(0.75)
                                if not ok:
                                    with np.errstate(all="ignore"):
                                        c = float(F(dot(nrm, n_local) / F(length(nrm) * length(n_local))))
                                    ok = (c <= 1.0) and abs(math.acos(c)) < 0.5 if -1.0 <= c <= 1.0 else False   # acos outside [-1, 1]: NaN
                                if dist < best_dist and ok:
                                    counter += 1
                                    best_dist, best = dist, cur
                        v = F(v + sy)
                    u = F(u + sx)
                if counter > 0:
                    update_id, t_last = 1, F(-1)
                else:
                    update_id, t_last = 2, F(-2)
                    conf = F(0.08) if prob > F(0.5) else F(0)
            if update_id > 0:                                              # data.geom:35
                vert = np.array([world[0], world[1], world[2], conf, color, F(1), F(time), t_last,
                                 n_world[0], n_world[1], n_world[2], radius_n], F)
                if update_id == 1:                                         # data.frag + the depth test: the first fragment at a texel stays
                    int_y = best // TEX_DIM
                    int_x = best - int_y * TEX_DIM
                    update_map.setdefault((int_x, int_y), len(emitted))
                emitted.append((vert, update_id, best))
    return emitted, update_map


def update_pass(surf, emitted, update_map, time):
    out = surf.copy()
    merged = []
    for s in range(surf.shape[0]):
        int_y = s // TEX_DIM
        int_x = s - int_y * TEX_DIM
        k = update_map.get((int_x, int_y))
        if k is None:
            continue                                                       # update.vert:100-106
        merged.append(s)
        new = emitted[k][0]
        q = surf[s]
        c_k, a = q[3], new[3]
        hist = q[5]
        a = max(F(0.01), min(F(0.53), F(F(F(2) * a) * a)))                 # :63
        c_k = max(F(0.01), min(c_k, F(0.99)))                              # :64
        ltm = F(math.log(float(F(F(F(1.0) / F(F(1.0) - c_k)) - F(1.0)))))  # :66
        ltm = F(ltm + F(math.log(float(F(a / F(F(1.0) - a))))))            # :67
        c_k1 = F(F(1.0) - F(F(1.0) / F(F(1.0) + F(math.exp(float(ltm))))))  # :68
        if new[11] < F(F(1.5) * q[11]):                                    # :70
            w, den = F(hist * c_k), F(F(hist * c_k) + a)
            mix = lambda o, n: F(F(F(w * o) + F(a * n)) / den)
            out[s, 0:3] = [mix(q[k2], new[k2]) for k2 in range(3)]
            out[s, 3] = c_k1
            oc, nc = decode_color(q[4]), decode_color(new[4])
            out[s, 4] = encode_color([mix(oc[k2], nc[k2]) for k2 in range(3)])
            out[s, 5], out[s, 6], out[s, 7] = F(hist + F(1)), q[6], F(time)
            nr = np.array([mix(q[8 + k2], new[8 + k2]) for k2 in range(4)], F)
            out[s, 8:11] = normalize(nr[:3])
            out[s, 11] = nr[3]
        else:                                                              # :85-98
            out[s, 3], out[s, 5], out[s, 7] = c_k1, F(hist + F(1)), F(time)
    return out, merged


def clean_pass(vertices, t_inv, cam, time, time_delta, conf_threshold, surf, index):
    """copy_unstable.vert/.geom over `vertices` (the merged model, then the new-unstable buffer); returns kept vertices + flags"""
    cx, cy, fx, fy = cam
    fc, fr_ = F(COLS), F(ROWS)
    kept, flags = [], []
    for q in vertices:
        out = q.copy()
        test = 1
        lp = mat4_point(t_inv, q[0:3])                                     # :46
        with np.errstate(all="ignore"):
            x = F(F(F(fx * lp[0]) / lp[2]) + cx)                           # :48-49
            y = F(F(F(fy * lp[1]) / lp[2]) + cy)
        scale, wm = F(FACTOR), F(2)
        sx, sy = F(F(F(1.0) / F(fc * scale)) * F(0.5)), F(F(F(1.0) / F(fr_ * scale)) * F(0.5))
        count = z_count = 0
        if F(F(time) - q[7]) < F(time_delta) and lp[2] > 0 and x > 0 and y > 0 and x < fc and y < fr_:   # :61
            u = F(F(x / fc) - F(F(scale * sx) * wm))
            while u < F(F(x / fc) + F(F(scale * sx) * wm)):
                v = F(F(y / fr_) - F(F(scale * sy) * wm))
                while v < F(F(y / fr_) + F(F(scale * sy) * wm)):
                    cur = int(fetch(index, u, v))
                    if cur > 0:
                        pos, conf, ct, _, _ = index_textures(surf, t_inv, cur)
                        ddx, ddy = F(pos[0] - lp[0]), F(pos[1] - lp[1])
                        if (ct[2] < q[6] and conf > F(conf_threshold) and pos[2] > lp[2] and F(pos[2] - lp[2]) < F(0.01)
                                and F(math.sqrt(F(F(ddx * ddx) + F(ddy * ddy)))) < F(q[11] * F(1.4))):   # :74-78
                            count += 1
                        if ct[3] == F(time) and conf > F(F(0.4) * F(conf_threshold)) and pos[2] > lp[2] and F(pos[2] - lp[2]) > F(0.01):  # :83-86
                            z_count += 1
                    v = F(v + sy)
                u = F(u + sx)
        if count > 6 or z_count > 5:                                       # :95
            test = 0
        if out[7] == F(-2):                                                # :101
            out[7] = F(time)
        if (out[7] == F(-1) or (F(F(time) - out[7]) > 10 and out[3] < F(0.5))) or out[3] == F(0):   # :108
            test = 0
        if out[7] > 0 and F(F(time) - out[7]) > F(time_delta):             # :113
            test = 1
        flags.append(test)
        if test > 0:
            kept.append(out)
    return kept, flags


def feedback_pass(depth_img, colour_of, cam, time, max_depth):
    """FeedbackBuffer::compute: vertex_feedback.vert + .geom over the uv buffer (x outer, y inner) -> emitted (pos, colour, normRad)"""
    cx, cy, fx, fy = cam
    camz, camw = F(F(1.0) / fx), F(F(1.0) / fy)                            # FeedbackBuffer.cpp:89-92 (float division)
    fc, fr_ = F(COLS), F(ROWS)
    out = []
    for i in range(COLS):
        for j in range(ROWS):
            tx = F(float(F(i) / fc) + 1.0 / (2 * float(fc)))              # FeedbackBuffer.cpp:46-47
            ty = F(float(F(j) / fr_) + 1.0 / (2 * float(fr_)))
            x, y = F(tx * fc), F(ty * fr_)

            def vertex(u, v, xx, yy):
                z = F(fetch(depth_img, u, v))
                return v3(F(F(F(xx - cx) * z) * camz), F(F(F(yy - cy) * z) * camw), z)

            one_c, one_r = F(F(1.0) / fc), F(F(1.0) / fr_)
            pos = vertex(tx, ty, x, y)
            xf, xb = vertex(F(tx + one_c), ty, F(x + F(1)), y), vertex(F(tx - one_c), ty, F(x - F(1)), y)
            yf, yb = vertex(tx, F(ty + one_r), x, F(y + F(1))), vertex(tx, F(ty - one_r), x, F(y - F(1)))
            half = lambda a, b: np.array([F(F(a[k] + b[k]) / F(2)) for k in range(3)], F)
            n = normalize(cross((half(xb, pos) - half(xf, pos)).astype(F), (half(yb, pos) - half(yf, pos)).astype(F)))
            mean_focal = F(F(F(F(1.0) / abs(camz)) + F(F(1.0) / abs(camw))) / F(2.0))
            radius = F(F(pos[2] / mean_focal) * F(1.41421356237))
            with np.errstate(all="ignore"):
                radius_n = min(F(F(2.0) * radius), F(radius / abs(n[2])))
            if pos[2] <= 0 or pos[2] > max_depth:                          # zVal = 0: vertex_feedback.geom emits nothing
                continue
            colour = np.array([encode_color(colour_of(tx, ty)), F(0), F(0), F(time)], F)   # vColor.x, .y = 0, .w = time (.z: the texture's blue)
            out.append((pos, colour, np.array([n[0], n[1], n[2], radius_n], F)))
    return out


def init_model(fr, pose, cam, time, max_depth):
    """GlobalModel::initialise: RAW feedback (position, colour) paired by index with FILTERED feedback (normal / radius, b as colour)"""
    raw = feedback_pass(fr["raw"], lambda u, v: fetch(fr["rgb"], u, v).astype(F) / F(255), cam, time, max_depth)
    fil = feedback_pass(fr["fil"], lambda u, v: np.full(3, fetch(fr["b"], u, v), F), cam, time, max_depth)   # a LUMINANCE texture reads (L, L, L)
    surf = np.zeros((len(raw), 12), F)
    for k, (pos, colour, _) in enumerate(raw):
        nr = fil[k][2] if k < len(fil) else np.zeros(4, F)                # the buffers start zero-filled
        prob = fil[k][1][0] if k < len(fil) else F(0)
        surf[k, 0:3] = mat4_point(pose, pos)                               # init_unstable.vert:36
        surf[k, 3] = decode_color(prob)[0]                                 # :38-40
        surf[k, 4], surf[k, 5], surf[k, 6], surf[k, 7] = colour[0], F(1), F(1), colour[3]   # :42-45
        surf[k, 8:11] = mat3_vec(pose, nr[0:3])                            # :47
        surf[k, 11] = nr[3]
    return surf


def splat_pass(surf, t_inv, cam, time, max_time, time_delta, max_depth, conf_threshold):
    """IndexMap::combinedPredict for one confidence level: vertex.z and the RGBA8 image of the nearest surfel per pixel.
    A point sprite covers the pixels whose centres lie within gl_PointSize / 2 of its centre."""
    cx, cy, fx, fy = cam
    fc, fr_ = F(COLS), F(ROWS)
    zimg = np.zeros((ROWS, COLS), F)
    rgb = np.zeros((ROWS, COLS, 3), np.uint8)
    zbuf = np.ones((ROWS, COLS), F)
    for s in range(surf.shape[0]):
        q = surf[s]
        h = mat4_point(t_inv, q[0:3])                                      # splat.vert:55
        if h[2] > max_depth or h[2] < F(0.4) or q[3] < F(conf_threshold) or F(F(time) - q[7]) > F(time_delta) or q[7] > F(max_time):  # :57
            continue
        with np.errstate(all="ignore"):
            ndc_x = F(F(F(F(F(fx * h[0]) / h[2]) + cx) - F(fc * F(0.5))) / F(fc * F(0.5)))   # projectPoint :39-44
            ndc_y = F(F(F(F(F(fy * h[1]) / h[2]) + cy) - F(fr_ * F(0.5))) / F(fr_ * F(0.5)))
        if not (-1 <= ndc_x <= 1 and -1 <= ndc_y <= 1):
            continue
        xw, yw = F(F(ndc_x + F(1)) * F(fc * F(0.5))), F(F(ndc_y + F(1)) * F(fr_ * F(0.5)))
        n = normalize(mat3_vec(t_inv, q[8:11]))                            # :68
        rad = q[11]
        t = normalize(v3(F(n[1] - n[2]), F(-n[0]), n[0]))
        x1 = np.array([F(F(t[k] * rad) * F(1.41421356)) for k in range(3)], F)   # :70
        y1 = cross(n, x1)                                                  # :72
        def proj(p):                                                       # projectPointImage :46-51
            with np.errstate(all="ignore"):
                return F(F(F(fx * p[0]) / p[2]) + cx), F(F(F(fy * p[1]) / p[2]) + cy)
        pts = [proj((h + x1).astype(F)), proj((h + y1).astype(F)), proj((h - y1).astype(F)), proj((h - x1).astype(F))]
        x_diff = abs(F(max(p_[0] for p_ in pts) - min(p_[0] for p_ in pts)))
        y_diff = abs(F(max(p_[1] for p_ in pts) - min(p_[1] for p_ in pts)))
        size = max(F(0), max(x_diff, y_diff))                              # :85
        if not size > 0:
            continue
        half = F(size * F(0.5))
        i0, i1 = max(0, int(math.ceil(F(F(xw - half) - F(0.5))))), min(COLS - 1, int(math.floor(F(F(xw + half) - F(0.5)))))
        j0, j1 = max(0, int(math.ceil(F(F(yw - half) - F(0.5))))), min(ROWS - 1, int(math.floor(F(F(yw + half) - F(0.5)))))
        pn = dot(h, n)
        for j in range(j0, j1 + 1):
            for i in range(i0, i1 + 1):
                fcx, fcy = F(F(i) + F(0.5)), F(F(j) + F(0.5))              # gl_FragCoord
                l = normalize(v3(F(F(fcx - cx) / fx), F(F(fcy - cy) / fy), F(1)))   # combo_splat.frag:37
                with np.errstate(all="ignore"):
                    k_ = F(pn / dot(l, n))
                corrected = np.array([F(k_ * l[0]), F(k_ * l[1]), F(k_ * l[2])], F)   # :39
                diff = (corrected - h).astype(F)
                if dot(diff, diff) > F(rad * rad):                         # :42-49
                    continue
                depth = F(F(corrected[2] / F(F(2) * max_depth)) + F(0.5))   # :63
                if not (0 <= depth <= 1) or not depth < zbuf[j, i]:
                    continue
                zbuf[j, i] = depth
                zimg[j, i] = corrected[2]
                col = int(q[4])
                rgb[j, i] = ((col >> 16) & 0xFF, (col >> 8) & 0xFF, col & 0xFF)
    return zimg, rgb


def predict_images(surf, pose, cam, time, time_delta, max_depth, conf_low, conf_high, extract_max, fr_prev):
    """Reconstruction::getPredictedImages: depthPrediction, intensityPrediction as (rows, cols) arrays"""
    t_inv = np.linalg.inv(pose.astype(np.float64)).astype(F)
    z_lo, rgb_lo = splat_pass(surf, t_inv, cam, time, time, time_delta, max_depth, conf_low)
    z_hi, rgb_hi = splat_pass(surf, t_inv, cam, time, time, time_delta, max_depth, conf_high)
    rw, rh = COLS // 40, ROWS // 40                                        # Resize to 1/40 + denseEnough
    dense = False
    if rw * rh > 0:
        hits = 0
        for j in range(rh):
            for i in range(rw):
                sx = min(COLS - 1, int(F(F(F(F(i) + F(0.5)) / F(rw)) * F(COLS))))
                sy = min(ROWS - 1, int(F(F(F(F(j) + F(0.5)) / F(rh)) * F(ROWS))))
                hits += int(all(rgb_lo[sy, sx] > 0))
        dense = F(hits) / F(rh * rw) > F(0.25)
    lo_empty = rgb_lo.astype(int).sum(2) == 0
    hi_empty = rgb_hi.astype(int).sum(2) == 0
    if not dense:                                                          # Reconstruction.cpp:663-669
        zr = (fr_prev["fil_mm"].astype(F) / F(1000.0)).astype(F)
        z1 = np.where(z_lo == 0, np.where(fr_prev["b"] > F(0.6), zr, F(0)), z_lo)      # fill_vertex.frag
        z = np.where(z_hi == 0, z1, z_hi)                                  # fill_vertex_from_texture.frag
        c1 = np.where(lo_empty[..., None], fr_prev["rgb"], rgb_lo)         # fill_rgb.frag
        c = np.where(hi_empty[..., None], c1, rgb_hi)
    else:                                                                  # :696-700
        z = np.where(z_hi == 0, z_lo, z_hi)
        c = np.where(hi_empty[..., None], rgb_lo, rgb_hi)
    depth = np.where((z > F(extract_max)) | (z <= 0), F(0), z).astype(F)   # extract_depth.frag
    nf = F(F(1.0) / F(255.0))
    r, g, b = [(c[..., k].astype(F) * nf).astype(F) for k in range(3)]
    inten = ((F(0.299) * r + F(0.587) * g).astype(F) + F(0.114) * b).astype(F)   # :692
    return depth, inten, bool(dense)


def weighting(last_pose, curr_pose, multiplier):
    from scipy.spatial.transform import Rotation

    diff = np.linalg.inv(curr_pose.astype(np.float64)) @ last_pose.astype(np.float64)
    w = max(np.linalg.norm(diff[:3, 3]), np.linalg.norm(Rotation.from_matrix(diff[:3, :3]).as_rotvec()))
    w = min(w, 0.15)
    return F(max(1.0 - w / 0.15, 0.5) * multiplier)


# ------------------------------------------------------------------------------------------------
def synth(k, rng):
    """full-resolution frame k of a small scene: a slanted wall with a bump that the camera approaches"""
    H, W = ROWS * RES, COLS * RES
    yy, xx = np.mgrid[0:H, 0:W]
    z = 1.6 + 0.004 * xx + 0.002 * yy - 0.03 * k
    z -= 0.25 * np.exp(-((xx - 40 - 1.5 * k) ** 2 + (yy - 30) ** 2) / 90.0)
    if k == 2:
        z[14:40, 18:44] = 1.05 + 0.001 * xx[14:40, 18:44]                  # a box appears in front of the wall ...
    if k == 3:
        z[14:40, 30:56] = 1.05 + 0.001 * xx[14:40, 30:56]                  # ... and has moved sideways in the next frame
    depth = np.clip(np.rint(z * 1000), 0, 65535).astype(np.uint16)
    depth[5:9, 60:66] = 0                                                  # a hole
    color = rng.integers(1, 256, (H, W, 3)).astype(np.uint8)
    return color, depth


def main():
    from oracle import binding
    import staticfusion_amd as sf
    from staticfusion_amd.synth import se3_exp

    rng = np.random.default_rng(2024)
    binding.build()
    ora = binding.load()
    p = ora.default_params_struct()
    p.ctf_levels = 2
    s = sf.Solver(ora, ROWS, COLS, 1, p)
    mp = s.default_model_params()
    cam = (F(mp.cx), F(mp.cy), F(mp.fx), F(mp.fy))
    yy, xx = np.mgrid[0:ROWS, 0:COLS]
    labels = (((xx // 5) + 8 * (yy // 5)) % 24).astype(np.int32)
    b_segm = np.linspace(0.2, 1.0, 24).astype(F)
    incs = [np.eye(4), se3_exp(np.array([0.004, -0.002, 0.03, 0.002, -0.003, 0.001])), se3_exp(np.array([-0.003, 0.002, 0.03, -0.001, 0.002, 0.002])),
            se3_exp(np.array([0.012, 0.004, 0.03, 0.004, -0.02, 0.003])), se3_exp(np.array([0.0, 0.0, 0.03, 0.0, 0.0, 0.0]))]
    out = dict(rows=ROWS, cols=COLS, res_factor=RES, labels=labels, b_segm=b_segm, increments=np.stack(incs).astype(F),
               conf_threshold=F(mp.conf_high), max_depth=F(mp.max_depth), time_delta=np.int64(mp.time_delta))
    pose = np.eye(4, dtype=F)
    surf = None
    for k in range(5):
        color_full, depth_full = synth(k, rng)
        out["color_full_%d" % k], out["depth_full_%d" % k] = color_full, depth_full
        # the independent input stage: decimate / flip, bilateral filter, metricise; buildSegmImage
        inten, depth_loaded, mm, color = load_frame(color_full, depth_full, RES)
        fil_mm = bilateral_mm(mm)
        fr = dict(raw=metricise(mm), fil=metricise(fil_mm), rgb=color, b=segm_image(labels, b_segm, np.ones(24, F)), fil_mm=fil_mm)
        if k == 0:
            # the oracle is used here only to cross-check the inputs the two sides must agree on before the fusion starts
            s.load_frame(0, color_full, depth_full, RES)
            s.filter_depth()
            s.set_segm_state(0, labels, b_segm, np.ones(24, np.float32))
            s.build_segm_image()
            assert np.array_equal(s.b_image(), fr["b"]) and np.array_equal(s.input_image(sf.capi.IN_DEPTH_METRIC), fr["raw"])
            assert np.array_equal(s.current()[0], fr["fil"]) and np.array_equal(s.input_image(sf.capi.IN_COLOR), fr["rgb"])
            surf = init_model(fr, pose, cam, 1, F(mp.max_depth))
            out["map_0"] = surf.copy()
            continue
        time = k + 1
        last = pose.copy()
        pose = (pose.astype(F) @ incs[k].astype(F)).astype(F)   # currPose * inPose in float (the order of the 4 products differs: tolerance)
        wgt = weighting(last, pose, 1.0)
        t_inv = np.linalg.inv(pose.astype(np.float64)).astype(F)
        index1 = predict_indices(surf, t_inv, cam, time, mp.time_delta, F(mp.max_depth))
        emitted, update_map = data_pass(fr, pose, t_inv, cam, time, wgt, F(mp.max_depth), surf, index1)
        merged_model, merged_ids = update_pass(surf, emitted, update_map, time)
        index2 = predict_indices(merged_model, t_inv, cam, time, mp.time_delta, F(mp.max_depth))
        verts = [merged_model[q] for q in range(merged_model.shape[0])] + [e[0] for e in emitted]
        kept, flags = clean_pass(verts, t_inv, cam, time, mp.time_delta, mp.conf_high, merged_model, index2)
        surf = np.stack(kept).astype(F)
        out["weighting_%d" % k] = wgt
        out["index_first_%d" % k], out["index_merged_%d" % k] = index1, index2
        out["update_id_%d" % k] = np.array([e[1] for e in emitted], np.int32)
        out["best_%d" % k] = np.array([e[2] for e in emitted], np.int64)
        out["merged_ids_%d" % k] = np.array(merged_ids, np.int64)
        out["keep_flags_%d" % k] = np.array(flags, np.int8)
        out["map_%d" % k] = surf.copy()
        print("frame %d: %d emitted, %d associated, %d surfels merged, %d -> %d surfels" %
              (k, len(emitted), int((out["update_id_%d" % k] == 1).sum()), len(merged_ids), merged_model.shape[0], surf.shape[0]))
    # Reconstruction::getPredictedImages from the final map at the final pose (tick = 6), fill-in from the last frame
    for name, lo, hi in (("", mp.conf_low, mp.conf_high), ("_strict", 0.6, 0.95)):  # the second one leaves holes for the fill-in
        d, i_, dense = predict_images(surf, pose, cam, 6, mp.time_delta, F(mp.max_depth), lo, hi, mp.extract_max_depth, fr)
        out["pred_depth" + name], out["pred_intensity" + name], out["pred_dense" + name] = d, i_, np.bool_(dense)
        print("prediction%s: %d pixels drawn, dense %s" % (name, int((d > 0).sum()), dense))
    path = os.path.join(ROOT, "tests", "golden", "fusion_40x30.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
