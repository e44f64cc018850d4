"""CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE) for the on-disk formats:
an independent Python/NumPy restatement of
  StaticFusion::loadAssoc              reference FrontEnd.cpp:183-214
  the PNG decoding cv::imread performs for FrontEnd.cpp:220,240 (PNG: ISO/IEC 15948; OpenCV absent here)
  Datasets::writeTrajectoryFile        reference Utils/Datasets.cpp:252-265 (+ rotateByZ :57-59)
  the pose log of Reconstruction       reference Reconstruction.cpp:53-81
PARITY UNPINNED: the reference holds no fixtures for these either. Only tests import this module."""
import math
import struct
import zlib

import numpy as np

f32 = np.float32


def load_assoc(directory, assoc_file):
    ts, fd, fc = [], [], []
    with open(directory + assoc_file) as f:  # :186 plain concatenation
        for line in f.read().split("\n"):
            if line == "" or line.startswith("#"):  # :199
                continue
            tok = line.split()
            try:  # iss >> double >> string >> double >> string (:204)
                tc, c, td, d = float(tok[0]), tok[1], float(tok[2]), tok[3]
            except (IndexError, ValueError):
                break
            ts.append(td); fd.append(directory + d); fc.append(directory + c)
    return ts, fd, fc


def _png_rows(data):
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, plte = 8, b"", None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert zlib.crc32(typ + body) == struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]
        if typ == b"IHDR":
            w, h, bd, ct, _, _, il = struct.unpack(">IIBBBBB", body)
            assert il == 0
        elif typ == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ct]
    bits = ch * bd
    stride = (w * bits + 7) // 8
    bpp = max(1, bits // 8)
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, stride + 1)
    out = np.zeros((h, stride), np.int64)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int64)
        up = out[y - 1] if y else np.zeros(stride, np.int64)
        cur = out[y]
        if ft == 0:
            cur[:] = line
        elif ft == 2:
            cur[:] = (line + up) & 255
        else:
            for x in range(stride):
                a = cur[x - bpp] if x >= bpp else 0
                b = up[x]
                c = up[x - bpp] if x >= bpp else 0
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) // 2
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + pred) & 255
    rows = out.astype(np.uint8)
    if bd == 16:
        s = rows.reshape(h, w * ch, 2).astype(np.uint16)
        samples = (s[..., 0] << 8) | s[..., 1]
    elif bd == 8:
        samples = rows.astype(np.uint16)
    else:
        per = 8 // bd
        bitsarr = np.unpackbits(rows, axis=1).reshape(h, stride * per, bd)
        samples = (bitsarr * (1 << np.arange(bd - 1, -1, -1))).sum(axis=2)[:, :w].astype(np.uint16)
    return samples.reshape(h, w, ch), bd, ct, plte


def decode_color(data):
    """what cv::imread(path, CV_LOAD_IMAGE_COLOR) returns: (rows, cols, 3) uint8, B G R"""
    s, bd, ct, plte = _png_rows(data)
    if ct == 3:
        rgb = plte[s[..., 0]]
    else:
        v = (s >> 8) if bd == 16 else ((s * 255 // ((1 << bd) - 1)) if (bd < 8 and ct == 0) else s)
        v = v.astype(np.uint8)
        rgb = np.repeat(v[..., :1], 3, axis=2) if ct in (0, 4) else v[..., :3]
    return np.ascontiguousarray(rgb[..., ::-1])


def decode_depth16(data):
    s, bd, ct, _ = _png_rows(data)
    assert ct == 0 and bd in (8, 16)
    return s[..., 0].astype(np.uint16)


def pose_compose(a, b):
    a, b = np.asarray(a, f32), np.asarray(b, f32)
    out = np.zeros((4, 4), f32)
    for i in range(4):
        for j in range(4):
            s = a[i, 0] * b[0, j]
            for k in range(1, 4):
                s = f32(s + a[i, k] * b[k, j])
            out[i, j] = s
    return out


def _quat(m):
    """Eigen::Quaternionf(Matrix3f) -> (x, y, z, w), float32"""
    t = f32(f32(m[0, 0] + m[1, 1]) + m[2, 2])
    q = np.zeros(4, f32)
    if t > 0:
        t = np.sqrt(f32(t + f32(1)))
        q[3] = f32(0.5) * t
        t = f32(0.5) / t
        q[0] = f32(m[2, 1] - m[1, 2]) * t
        q[1] = f32(m[0, 2] - m[2, 0]) * t
        q[2] = f32(m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(f32(f32(f32(m[i, i] - m[j, j]) - m[k, k]) + f32(1)))
        q[i] = f32(0.5) * t
        t = f32(0.5) / t
        q[3] = f32(m[k, j] - m[j, k]) * t
        q[j] = f32(m[j, i] + m[i, j]) * t
        q[k] = f32(m[k, i] + m[i, k]) * t
    return q


def _g(x):  # std::ostream << float with default precision 6
    return "%g" % float(x)


def trajectory_line(timestamp, pose, rotate_by_z):
    pose = np.asarray(pose, f32)
    if rotate_by_z:
        s, c = f32(math.sin(float(f32(math.pi)))), f32(math.cos(float(f32(math.pi))))
        rz = np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, f32(f32(f32(1) - c) + c), 0], [0, 0, 0, 1]], f32)
        pose = pose_compose(pose, rz)
        ts = "%.04f" % timestamp
    else:
        ts = "%.6f" % timestamp
    q = _quat(pose)
    return " ".join([ts, _g(pose[0, 3]), _g(pose[1, 3]), _g(pose[2, 3]), _g(q[0]), _g(q[1]), _g(q[2]), _g(q[3])]) + "\n"


def save_ply_bytes(surfels, conf_threshold):
    """Reconstruction::savePly (reference Reconstruction.cpp:358-455): the bytes of the .ply it writes"""
    s = np.asarray(surfels, f32).reshape(-1, 12)
    keep = s[s[:, 3] > f32(conf_threshold)]
    head = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z"
            "\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nproperty float nx\nproperty float ny\nproperty float nz"
            "\nproperty float radius\nend_header\n" % keep.shape[0]).encode()
    rec = np.zeros(keep.shape[0], np.dtype([("p", "<f4", 3), ("c", "u1", 3), ("n", "<f4", 3), ("r", "<f4")]))
    rec["p"] = keep[:, 0:3]
    col = keep[:, 4].astype(np.int64)
    rec["c"] = np.stack([(col >> 16) & 0xFF, (col >> 8) & 0xFF, col & 0xFF], 1)
    rec["n"] = keep[:, 8:11] * f32(-1)
    rec["r"] = keep[:, 11]
    return head + rec.tobytes()
