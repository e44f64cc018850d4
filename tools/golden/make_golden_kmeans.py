"""Independent Python (float32 scalars) derivation of the geometric clustering, used to emit
tests/golden/kmeans_160x120.npz: the labels are what BASELINE.json asks to be BIT-EXACT.

Written from the reference's KMeans.cpp, not from the C++ oracle or the HIP kernels, and sharing no code with them: pixel by
pixel in the reference's loop order, one float32 operation at a time.

  initialise        StaticFusion::initializeKMeans                  reference KMeans.cpp:64-135
  lloyd             StaticFusion::kMeans3DCoord, the iterations     KMeans.cpp:137-233
  label_level0      ... the labelling at the maximum resolution     KMeans.cpp:236-288
  connectivity      StaticFusion::computeRegionConnectivity         KMeans.cpp:296-341
  label_pyramid     StaticFusion::createClustersPyramidUsingKMeans  KMeans.cpp:343-391

Two things the source does not determine are taken as DESIGN.md section 6 states them: squaredNorm() of a 3-vector is
((a^2 + b^2) + c^2), and std::sort's order of EQUAL centre distances is by index (the reference's std::sort is unstable).
The depth pyramid comes from the independent derivation of createImagePyramid (make_golden.py).

Run (in the build container):  python tools/golden/make_golden_kmeans.py  -> tests/golden/kmeans_160x120.npz
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from make_golden import pyramid_level, tan_half_fovh, xx_yy  # noqa: E402

F = np.float32
NC = 24


def c_round(x):  # C round(): halves away from zero
    return int(math.floor(float(x) + 0.5)) if x >= 0 else -int(math.floor(-float(x) + 0.5))


def sqnorm(a, b):  # (a - b).squaredNorm()
    d0, d1, d2 = F(a[0] - b[0]), F(a[1] - b[1]), F(a[2] - b[2])
    return F(F(F(d0 * d0) + F(d1 * d1)) + F(d2 * d2))


def sorted_distances(centres):
    """per label l: the other labels ordered by their squared distance to l (KMeans.cpp:172-183)"""
    out = []
    for l in range(NC):
        d = [sqnorm(centres[:, l], centres[:, li]) for li in range(NC)]
        order = sorted(range(NC), key=lambda li: (d[li], li))
        out.append([(li, d[li]) for li in order])
    return out


def search(p, last, centres, dists):
    """the pruned nearest-centre search (KMeans.cpp:196-212)"""
    best = last
    d_last = sqnorm(centres[:, last], p)
    best_d = d_last
    for li in range(1, NC):
        idx, dist = dists[last][li]
        if dist > F(F(4.0) * d_last):
            break
        d = sqnorm(centres[:, idx], p)
        if d < best_d:
            best_d, best = d, idx
    return best


def initialise(depth1):
    rows_km, cols_km = depth1.shape
    labels = np.full((rows_km, cols_km), NC, np.int32)
    vert_div = int(math.ceil(math.sqrt(NC)))
    u_div = F(F(cols_km) / F(NC + 1))
    v_div = F(F(rows_km) / F(vert_div + 1))
    u_label = [c_round(F(F(i + 1) * u_div)) for i in range(NC)]
    v_label = [c_round(F(F(i % vert_div + 1) * v_div)) for i in range(NC)]
    for u in range(cols_km):
        for v in range(rows_km):
            if depth1[v, u] != 0:
                min_dist, ini = 1000000, NC
                for l in range(NC):
                    q = (v - v_label[l]) ** 2 + (u - u_label[l]) ** 2
                    if q < min_dist:
                        ini, min_dist = l, q
                labels[v, u] = ini
    members = [[] for _ in range(NC)]
    for u in range(cols_km):
        for v in range(rows_km):
            if depth1[v, u] != 0:
                members[labels[v, u]].append(depth1[v, u])
    inv_f = F(F(F(2.0) * tan_half_fovh()) / F(cols_km))
    disp_u, disp_v = F(F(0.5) * F(cols_km - 1)), F(F(0.5) * F(rows_km - 1))
    centres = np.zeros((3, NC), F)
    for l in range(NC):
        if members[l]:
            med = sorted(members[l])[len(members[l]) // 2]             # nth_element at size / 2
            centres[0, l] = med
            centres[1, l] = F(F(F(F(u_label[l]) - disp_u) * med) * inv_f)
            centres[2, l] = F(F(F(F(v_label[l]) - disp_v) * med) * inv_f)
    return labels, centres


def lloyd(depth1, xx1, yy1, labels, centres_a):
    rows_km, cols_km = depth1.shape
    iters = 0
    for _ in range(9):                                                 # iter_kmeans - 1
        iters += 1
        centres_b = np.zeros((3, NC), F)
        count = [0] * NC
        dists = sorted_distances(centres_a)
        for u in range(cols_km):
            for v in range(rows_km):
                if depth1[v, u] != 0:
                    p = (depth1[v, u], xx1[v, u], yy1[v, u])
                    best = search(p, int(labels[v, u]), centres_a, dists)
                    labels[v, u] = best
                    for r in range(3):
                        centres_b[r, best] = F(centres_b[r, best] + p[r])
                    count[best] += 1
        for l in range(NC):
            if count[l] > 0:
                for r in range(3):
                    centres_b[r, l] = F(centres_b[r, l] / F(count[l]))
        max_diff = max(abs(F(centres_a[r, l] - centres_b[r, l])) for r in range(3) for l in range(NC))
        centres_a = centres_b
        if max_diff < F(1e-2):
            break
    return labels, centres_a, iters


def label_level0(depth0, xx0, yy0, labels1, centres):
    rows, cols = depth0.shape
    labels0 = np.full((rows, cols), NC, np.int32)
    dists = sorted_distances(centres)
    for u in range(cols):
        for v in range(rows):
            if depth0[v, u] != 0:
                low = int(labels1[v // 2, u // 2])
                last = 0 if low == NC else low
                labels0[v, u] = search((depth0[v, u], xx0[v, u], yy0[v, u]), last, centres, dists)
    return labels0


def connectivity(depth0, xx0, yy0, labels0):
    rows, cols = depth0.shape
    thr = F(F(F(F(0.03) * F(120.0)) / F(rows)) * F(F(F(0.03) * F(120.0)) / F(rows)))
    conn = np.eye(NC, dtype=bool)
    sq = lambda a: F(a * a)
    for u in range(cols - 1):
        for v in range(rows - 1):
            if depth0[v, u] != 0:
                a = labels0[v, u]
                b = labels0[v + 1, u]
                if a != b and b != NC:
                    if F(sq(F(depth0[v, u] - depth0[v + 1, u])) + sq(F(yy0[v, u] - yy0[v + 1, u]))) < thr:
                        conn[a, b] = conn[b, a] = True
                b = labels0[v, u + 1]
                if a != b and b != NC:
                    if F(sq(F(depth0[v, u] - depth0[v, u + 1])) + sq(F(xx0[v, u] - xx0[v, u + 1]))) < thr:
                        conn[a, b] = conn[b, a] = True
    return conn


def label_pyramid(depth_l, xx_l, yy_l, centres):
    rows, cols = depth_l.shape
    labels = np.full((rows, cols), NC, np.int32)
    kd = {(la, lb): sqnorm(centres[:, la], centres[:, lb]) for la in range(NC) for lb in range(la + 1, NC)}
    for u in range(cols):
        for v in range(rows):
            if depth_l[v, u] != 0:
                p = (depth_l[v, u], xx_l[v, u], yy_l[v, u])
                label = 0
                min_dist = sqnorm(centres[:, 0], p)
                for l in range(1, NC):
                    if kd[(label, l)] > F(F(4.0) * min_dist):
                        continue
                    d = sqnorm(centres[:, l], p)
                    if d < min_dist:
                        label, min_dist = l, d
                labels[v, u] = label
    return labels


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "pyramid_160x120.npz"))
    depth = [g["d_new0"]]                     # 160 x 120 with a sphere, an invalid block and an invalid strip
    inten = [g["i_new0"]]
    for L in (1, 2, 3):
        d, i = pyramid_level(depth[-1], inten[-1])
        depth.append(d)
        inten.append(i)
    xy = [xx_yy(d) for d in depth]
    labels1, centres0 = initialise(depth[1])
    init_labels1 = labels1.copy()
    labels1, centres, iters = lloyd(depth[1], xy[1][0], xy[1][1], labels1, centres0.copy())
    labels0 = label_level0(depth[0], xy[0][0], xy[0][1], labels1, centres)
    conn = connectivity(depth[0], xy[0][0], xy[0][1], labels0)
    labels2 = label_pyramid(depth[2], xy[2][0], xy[2][1], centres)
    labels3 = label_pyramid(depth[3], xy[3][0], xy[3][1], centres)
    out = dict(depth0=depth[0], intensity0=inten[0], init_labels1=init_labels1, init_centres=centres0, labels0=labels0, labels1=labels1,
               labels2=labels2, labels3=labels3, centres=centres, connectivity=conn, iterations=np.int32(iters))
    path = os.path.join(ROOT, "tests", "golden", "kmeans_160x120.npz")
    np.savez_compressed(path, **out)
    print("k-means: %d iterations, cluster sizes at level 1: %s" % (iters, np.bincount(labels1.ravel(), minlength=25).tolist()))
    print("connected pairs: %d; wrote %s" % (int((conn.sum() - NC) // 2), path))


if __name__ == "__main__":
    main()
