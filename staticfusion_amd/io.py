"""ctypes binding of libsf_io.so (include/sf_io.h): association file, PNG frames, trajectory lines.
Host-only plumbing; nothing here computes on the GPU."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("SF_IO_LIB", os.path.join(_HERE, "csrc", "libsf_io.so"))

SIGNATURES = {
    "sf_io_assoc_load": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "sf_io_assoc_count": (C.c_int, [C.c_void_p]),
    "sf_io_assoc_entry": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]),
    "sf_io_assoc_free": (None, [C.c_void_p]),
    "sf_io_imread_color": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sf_io_imread_depth16": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sf_io_decode_color": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sf_io_decode_depth16": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sf_io_free": (None, [C.c_void_p]),
    "sf_io_pose_compose": (None, [C.POINTER(C.c_float)] * 3),
    "sf_io_trajectory_line": (C.c_int, [C.c_double, C.POINTER(C.c_float), C.c_int, C.c_char_p, C.c_size_t]),
    "sf_io_save_ply": (C.c_int, [C.c_char_p, C.POINTER(C.c_float), C.c_int, C.c_float]),
    "sf_io_last_error": (C.c_char_p, []),
}


class SfIoError(RuntimeError):
    pass


class Io:
    def __init__(self, lib_path=LIB):
        self.lib = C.CDLL(lib_path)  # raises OSError if the library was not built
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.lib, name)
            fn.restype, fn.argtypes = res, args

    def _check(self, rc):
        if rc < 0:
            raise SfIoError("libsf_io: %d: %s" % (rc, (self.lib.sf_io_last_error() or b"").decode()))
        return rc

    def load_assoc(self, directory, assoc_file="rgbd_assoc.txt"):
        """-> (timestamps, depth paths, colour paths), reference StaticFusion::loadAssoc"""
        h = C.c_void_p()
        self._check(self.lib.sf_io_assoc_load(directory.encode(), assoc_file.encode(), C.byref(h)))
        ts, fd, fc = [], [], []
        try:
            for i in range(self.lib.sf_io_assoc_count(h)):
                t, d, c = C.c_double(), C.c_char_p(), C.c_char_p()
                self._check(self.lib.sf_io_assoc_entry(h, i, C.byref(t), C.byref(d), C.byref(c)))
                ts.append(t.value); fd.append(d.value.decode()); fc.append(c.value.decode())
        finally:
            self.lib.sf_io_assoc_free(h)
        return ts, fd, fc

    def _image(self, call, dtype, channels):
        p, r, c = C.c_void_p(), C.c_int(), C.c_int()
        self._check(call(C.byref(p), C.byref(r), C.byref(c)))
        try:
            n = r.value * c.value * channels
            arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8 if dtype == np.uint8 else C.c_uint16)), shape=(n,)).copy()
        finally:
            self.lib.sf_io_free(p)
        return arr.reshape((r.value, c.value, 3) if channels == 3 else (r.value, c.value))

    def imread_color(self, path):
        return self._image(lambda *o: self.lib.sf_io_imread_color(path.encode(), *o), np.uint8, 3)

    def imread_depth16(self, path):
        return self._image(lambda *o: self.lib.sf_io_imread_depth16(path.encode(), *o), np.uint16, 1)

    def decode_color(self, data):
        return self._image(lambda *o: self.lib.sf_io_decode_color(data, len(data), *o), np.uint8, 3)

    def decode_depth16(self, data):
        return self._image(lambda *o: self.lib.sf_io_decode_depth16(data, len(data), *o), np.uint16, 1)

    def pose_compose(self, pose, T):
        """4x4 numpy (row, col) arrays -> pose @ T in float32 with the library's accumulation order"""
        a = np.ascontiguousarray(np.asarray(pose, np.float32).T)  # column-major storage
        b = np.ascontiguousarray(np.asarray(T, np.float32).T)
        out = np.zeros(16, np.float32)
        fp = C.POINTER(C.c_float)
        self.lib.sf_io_pose_compose(a.ctypes.data_as(fp), b.ctypes.data_as(fp), out.ctypes.data_as(fp))
        return out.reshape(4, 4).T.copy()

    def trajectory_line(self, timestamp, pose, rotate_by_z):
        a = np.ascontiguousarray(np.asarray(pose, np.float32).T)
        buf = C.create_string_buffer(256)
        self._check(self.lib.sf_io_trajectory_line(timestamp, a.ctypes.data_as(C.POINTER(C.c_float)), int(rotate_by_z), buf, 256))
        return buf.value.decode()

    def save_ply(self, path, surfels, conf_threshold):
        """Reconstruction::savePly's point cloud: the surfels above conf_threshold; returns the vertex count"""
        s = np.ascontiguousarray(surfels, dtype=np.float32).reshape(-1, 12)
        return self._check(self.lib.sf_io_save_ply(os.fsencode(path), s.ctypes.data_as(C.POINTER(C.c_float)), s.shape[0], conf_threshold))
