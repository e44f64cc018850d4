"""ctypes plumbing for the C ABI declared in include/sf.h.

`Api(lib_path, prefix)` binds one shared library implementing the ABI; `Solver` is a thin
numpy-in / numpy-out convenience wrapper over one `sf_handle`.  The product library is bound by
`staticfusion_amd.load()` (prefix ``sf_``); the test oracle binds the same declarations with the
prefix ``sfo_`` from `oracle/binding.py`.  Nothing here computes anything.
"""
import ctypes as C

import numpy as np

NUM_CLUSTERS = 24
MAX_OUTER = 32
HISTORY = 5

SET_NEW, SET_PRED, SET_WARPED, SET_INTER = 0, 1, 2, 3
CH_DEPTH, CH_INTENSITY, CH_XX, CH_YY = 0, 1, 2, 3
(LIN_DCU, LIN_DCV, LIN_DCT, LIN_DDU, LIN_DDV, LIN_DDT, LIN_WC, LIN_WD, LIN_NULL) = range(9)

IN_DEPTH_MM, IN_DEPTH_FILTERED_MM, IN_DEPTH_METRIC, IN_COLOR = range(4)

VARIANT_AUTO, VARIANT_THROUGHPUT, VARIANT_LATENCY, VARIANT_CLUSTER = range(4)
VARIANT_NAMES = {"auto": 0, "throughput": 1, "latency": 2, "cluster": 3}

STATUS_EIG_SKIPPED = 1
STATUS_EMPTY_LEVEL = 2
STATUS_SYNC_TIMEOUT = 4  # cluster build: a rendezvous timed out; the frame is not valid and the stream keeps its previous state


class SfParams(C.Structure):
    _fields_ = [
        ("ctf_levels", C.c_int32),
        ("max_iter_per_level", C.c_int32),
        ("max_iter_irls", C.c_int32),
        ("use_motion_filter", C.c_int32),
        ("segmentation_enabled", C.c_int32),
        ("debug_planes", C.c_int32),
        ("fovh", C.c_float),
        ("k_photometric_res", C.c_float),
        ("irls_delta_threshold", C.c_float),
        ("previous_speed_const_weight", C.c_float),
        ("previous_speed_eig_weight", C.c_float),
        ("kc_Cauchy", C.c_float),
        ("kb", C.c_float),
        ("kz", C.c_float),
        ("lambda_reg", C.c_float),
        ("lambda_prior", C.c_float),
    ]


class SfModelParams(C.Structure):
    _fields_ = [
        ("cx", C.c_float), ("cy", C.c_float), ("fx", C.c_float), ("fy", C.c_float),
        ("max_depth", C.c_float), ("conf_low", C.c_float), ("conf_high", C.c_float),
        ("time", C.c_int32), ("max_time", C.c_int32), ("time_delta", C.c_int32),
        ("extract_max_depth", C.c_float),
    ]


class SfOuterTrace(C.Structure):
    _fields_ = [
        ("level", C.c_int32),
        ("k", C.c_int32),
        ("n_valid", C.c_int32),
        ("irls_iters", C.c_int32),
        ("aver_res", C.c_float),
        ("var", C.c_float * 6),
        ("twist_level", C.c_float * 6),
        ("b_segm", C.c_float * NUM_CLUSTERS),
        ("T", C.c_float * 16),
        ("b_prior", C.c_float * NUM_CLUSTERS),
        ("lambda_t_w", C.c_float * NUM_CLUSTERS),
        ("AtA", C.c_float * 36),
        ("AtB", C.c_float * 6),
        ("delta_sol_max", C.c_float),
    ]


class SfFrameStats(C.Structure):
    _fields_ = [
        ("n_outer", C.c_int32),
        ("n_irls", C.c_int32),
        ("pixel_iters", C.c_int64),
        ("kmeans_iters", C.c_int32),
        ("status", C.c_int32),
        ("outer", SfOuterTrace * MAX_OUTER),
    ]


ABI_VERSION = 5     # SF_ABI_VERSION of include/sf.h (tests/test_capi_and_host.py holds the two together)
PROFILE_SLOTS = 32  # sf_get_stage_profile's array

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_H = C.c_void_p

# name -> (restype, argtypes); every symbol include/sf.h declares
SIGNATURES = {
    "default_params": (None, [C.POINTER(SfParams)]),
    "ctor_params": (None, [C.POINTER(SfParams)]),
    "create": (C.c_int, [C.POINTER(SfParams), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_H)]),
    "create_ex": (C.c_int, [C.POINTER(SfParams), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_H)]),
    "get_variant": (C.c_int, [_H, _ip, _ip, _ip]),
    "get_resident_workgroups": (C.c_int, [_H, _ip, _ip]),
    "destroy": (None, [_H]),
    "set_params": (C.c_int, [_H, C.POINTER(SfParams)]),
    "get_params": (C.c_int, [_H, C.POINTER(SfParams)]),
    "set_kb": (C.c_int, [_H, C.c_int, C.c_float]),
    "set_hip_stream": (C.c_int, [_H, C.c_void_p]),
    "synchronize": (C.c_int, [_H]),
    "last_error": (C.c_char_p, []),
    "backend": (C.c_char_p, []),
    "set_current": (C.c_int, [_H, C.c_int, _fp, _fp]),
    "set_prediction": (C.c_int, [_H, C.c_int, _fp, _fp]),
    "set_current_device": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "set_prediction_device": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "advance_sequences_device": (C.c_int, [_H, C.c_void_p, C.c_void_p, _ip, C.c_int]),
    "process_frames": (C.c_int, [_H, C.c_int, C.c_int, _fp]),
    "process_sequence_frames_device": (C.c_int, [_H, C.c_void_p, C.c_void_p, _ip, C.c_int, C.c_int, C.c_int, _fp]),
    "upload_current_async": (C.c_int, [_H, _fp, _fp]),
    "commit_upload": (C.c_int, [_H]),
    "alloc_pinned": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "free_pinned": (C.c_int, [C.c_void_p]),
    "current_to_prediction": (C.c_int, [_H]),
    "set_segm_state": (C.c_int, [_H, C.c_int, _ip, _fp, _fp]),
    "set_twist_old": (C.c_int, [_H, C.c_int, _fp]),
    "load_frame": (C.c_int, [_H, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.c_int, C.c_int, C.c_int]),
    "load_frame_device": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "set_depth_cutoff": (C.c_int, [_H, C.c_float]),
    "filter_depth": (C.c_int, [_H]),
    "get_current": (C.c_int, [_H, C.c_int, _fp, _fp]),
    "get_input_image": (C.c_int, [_H, C.c_int, C.c_int, C.c_void_p]),
    "timed_input_stage": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "default_model_params": (C.c_int, [_H, C.POINTER(SfModelParams)]),
    "predict_from_model": (C.c_int, [_H, C.c_int, _fp, C.c_int, _fp, C.POINTER(SfModelParams)]),
    "predict_from_model_device": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_int, _fp, C.POINTER(SfModelParams)]),
    "init_model_from_frame": (C.c_int, [_H, C.c_int, _fp, C.POINTER(SfModelParams), C.c_int, _fp, _ip]),
    "get_prediction": (C.c_int, [_H, C.c_int, _fp, _fp]),
    "get_prediction_dense": (C.c_int, [_H, _ip]),
    "get_prediction_dense_stream": (C.c_int, [_H, C.c_int, _ip]),
    "map_create": (C.c_int, [_H, C.c_int, C.POINTER(C.c_void_p)]),
    "map_destroy": (None, [C.c_void_p]),
    "map_fuse_frame": (C.c_int, [_H, C.c_int, C.c_void_p, _fp, C.c_float, C.POINTER(SfModelParams)]),
    "map_predict": (C.c_int, [_H, C.c_int, C.c_void_p, C.POINTER(SfModelParams)]),
    "map_fuse_frames": (C.c_int, [_H, C.c_int, _ip, C.POINTER(C.c_void_p), _fp, C.c_float, C.POINTER(SfModelParams)]),
    "map_predict_frames": (C.c_int, [_H, C.c_int, _ip, C.POINTER(C.c_void_p), C.POINTER(SfModelParams)]),
    "map_info": (C.c_int, [C.c_void_p, _ip, _ip, _fp, _ip]),
    "map_download": (C.c_int, [C.c_void_p, _fp, C.c_int]),
    "map_upload": (C.c_int, [C.c_void_p, _fp, C.c_int, _fp, C.c_int]),
    "map_get_index_map": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "build_pyramid": (C.c_int, [_H, C.c_int]),
    "kmeans": (C.c_int, [_H]),
    "run_solver": (C.c_int, [_H, C.c_int]),
    "push_history": (C.c_int, [_H, C.c_int]),
    "residuals_vs_history": (C.c_int, [_H, C.c_int]),
    "build_segm_image": (C.c_int, [_H]),
    "process_frame": (C.c_int, [_H, C.c_int]),
    "get_T": (C.c_int, [_H, C.c_int, _fp]),
    "get_twist": (C.c_int, [_H, C.c_int, _fp]),
    "get_twist_old": (C.c_int, [_H, C.c_int, _fp]),
    "get_b": (C.c_int, [_H, C.c_int, _fp]),
    "get_b_image": (C.c_int, [_H, C.c_int, _fp]),
    "get_labels": (C.c_int, [_H, C.c_int, C.c_int, _ip]),
    "get_kmeans": (C.c_int, [_H, C.c_int, _fp]),
    "get_connectivity": (C.c_int, [_H, C.c_int, C.POINTER(C.c_uint8)]),
    "get_cluster_residuals": (C.c_int, [_H, C.c_int, _fp]),
    "get_stats": (C.c_int, [_H, C.c_int, C.POINTER(SfFrameStats)]),
    "get_batch_results": (C.c_int, [_H, _fp, _ip, _ip, C.POINTER(C.c_int64)]),
    "get_plane": (C.c_int, [_H, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "get_lin_plane": (C.c_int, [_H, C.c_int, C.c_int, _fp, _ip, _ip]),
    "get_jacobian_rows": (C.c_int, [_H, C.c_int, _fp, _fp, _ip]),
    "level_rows": (C.c_int, [_H, C.c_int]),
    "level_cols": (C.c_int, [_H, C.c_int]),
    "batch": (C.c_int, [_H]),
    "timed_process_frames": (C.c_int, [_H, C.c_int, C.c_int, _fp]),
    "get_counters": (C.c_int, [_H] + [C.POINTER(C.c_int64)] * 4),
    "get_stage_profile": (C.c_int, [_H, C.POINTER(C.c_int64)]),
    "microbench_pass": (C.c_int, [_H, C.c_int, C.c_int, C.c_int, _fp]),
    "last_solver_kernel_ms": (C.c_int, [_H, _fp]),
    "clear_sync_timeout": (C.c_int, [_H]),
    "debug_stall_rank": (C.c_int, [_H, C.c_int, C.c_float, C.c_uint]),
    "microbench_copy": (C.c_int, [_H, C.c_size_t, C.c_int, _fp]),
    "abi_version": (C.c_int, [_ip, _ip, _ip]),
}


class SfError(RuntimeError):
    pass


class Api:
    """One loaded implementation of include/sf.h."""

    def __init__(self, lib_path, prefix):
        self.lib_path = str(lib_path)
        self.prefix = prefix
        self.lib = C.CDLL(self.lib_path)  # raises OSError if missing: callers must not swallow it
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.lib, prefix + name)  # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)
        # the library must have been built from the header these ctypes mirrors restate (silent struct drift is the likeliest
        # way for this binding to go wrong: a buffer that is too small is written past its end)
        sp, ss, slots = C.c_int(), C.c_int(), C.c_int()
        version = self.abi_version(C.byref(sp), C.byref(ss), C.byref(slots))
        if (version, sp.value, ss.value, slots.value) != (ABI_VERSION, C.sizeof(SfParams), C.sizeof(SfFrameStats), PROFILE_SLOTS):
            raise SfError("%s: ABI mismatch: the library reports version %d, sizeof(sf_params) %d, sizeof(sf_frame_stats) %d, %d profile "
                          "slots; this binding mirrors version %d, %d, %d, %d (rebuild: make -C staticfusion_amd/csrc && make -C oracle)"
                          % (self.lib_path, version, sp.value, ss.value, slots.value, ABI_VERSION, C.sizeof(SfParams), C.sizeof(SfFrameStats), PROFILE_SLOTS))

    def check(self, code):
        if code != 0:
            msg = self.last_error()
            raise SfError("%s%s failed with %d: %s" % (self.prefix, "call", code, msg.decode() if msg else ""))

    def default_params_struct(self):
        p = SfParams()
        self.default_params(C.byref(p))
        return p

    def ctor_params_struct(self):
        p = SfParams()
        self.ctor_params(C.byref(p))
        return p

    def backend_name(self):
        return self.backend().decode()

    def with_variant(self, variant):
        """A view of this binding whose Solver()s default to the named frame-kernel build (tests run every case on each)."""
        import copy

        v = copy.copy(self)
        v.default_variant = variant
        return v


def _f32(a):
    """column-major float32 contiguous buffer of a (rows, cols) array"""
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).T)


class Solver:
    """numpy convenience wrapper over one sf_handle (a batch of independent streams)."""

    def __init__(self, api, rows, cols, batch=1, params=None, device=0, variant=None):
        """variant: None = the api's default_variant (normally "auto": sf_create's own choice by batch size), or one of
        "auto" / "throughput" / "latency" / "cluster" (sf_create_ex)."""
        self.api = api
        self.rows, self.cols, self.batch_size = rows, cols, batch
        p = params if params is not None else api.default_params_struct()
        self.h = _H()
        v = variant if variant is not None else getattr(api, "default_variant", "auto")
        v = VARIANT_NAMES[v] if isinstance(v, str) else int(v)
        api.check(api.create_ex(C.byref(p), rows, cols, batch, device, v, C.byref(self.h)))
        self.params = SfParams()
        api.check(api.get_params(self.h, C.byref(self.params)))
        self.levels = self.params.ctf_levels

    def close(self):
        if self.h:
            self.api.destroy(self.h)
            self.h = _H()

    def variant(self):
        """(name, threads per workgroup, workgroups per stream) of the frame-kernel build this handle runs"""
        v, t, g = C.c_int32(), C.c_int32(), C.c_int32()
        self.api.check(self.api.get_variant(self.h, C.byref(v), C.byref(t), C.byref(g)))
        return {n: k for k, n in VARIANT_NAMES.items()}[v.value], t.value, g.value

    def resident_workgroups(self):
        """(workgroups per CU, launch grid) of the next frame launch"""
        p, t = C.c_int32(), C.c_int32()
        self.api.check(self.api.get_resident_workgroups(self.h, C.byref(p), C.byref(t)))
        return p.value, t.value

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- inputs -------------------------------------------------------------------------------
    def set_params(self, p):
        self.api.check(self.api.set_params(self.h, C.byref(p)))
        self.api.check(self.api.get_params(self.h, C.byref(self.params)))
        self.levels = self.params.ctf_levels

    def set_kb(self, kb, stream=-1):
        self.api.check(self.api.set_kb(self.h, stream, kb))

    def set_current(self, stream, depth, intensity):
        d, i = _f32(depth), _f32(intensity)
        assert d.shape == (self.cols, self.rows) and i.shape == d.shape
        self.api.check(self.api.set_current(self.h, stream, d.ctypes.data_as(_fp), i.ctypes.data_as(_fp)))

    def set_prediction(self, stream, depth, intensity):
        d, i = _f32(depth), _f32(intensity)
        assert d.shape == (self.cols, self.rows) and i.shape == d.shape
        self.api.check(self.api.set_prediction(self.h, stream, d.ctypes.data_as(_fp), i.ctypes.data_as(_fp)))

    # -- input stage ---------------------------------------------------
    def load_frame(self, stream, color_full, depth_full, res_factor=2):
        """color_full: (H, W, 3) uint8 in decoder order; depth_full: (H, W) uint16 millimetres."""
        cf = np.ascontiguousarray(color_full, dtype=np.uint8)
        df = np.ascontiguousarray(depth_full, dtype=np.uint16)
        assert cf.ndim == 3 and cf.shape[2] == 3 and df.shape == cf.shape[:2]
        self.api.check(self.api.load_frame(self.h, stream, cf.ctypes.data_as(C.POINTER(C.c_uint8)),
                                           df.ctypes.data_as(C.POINTER(C.c_uint16)), df.shape[0], df.shape[1], res_factor))

    def set_depth_cutoff(self, metres):
        self.api.check(self.api.set_depth_cutoff(self.h, metres))

    def filter_depth(self):
        self.api.check(self.api.filter_depth(self.h))

    def current(self, stream=0):
        """(depthCurrent, intensityCurrent) as (rows, cols) arrays"""
        d = np.zeros((self.cols, self.rows), dtype=np.float32)
        i = np.zeros((self.cols, self.rows), dtype=np.float32)
        self.api.check(self.api.get_current(self.h, stream, d.ctypes.data_as(_fp), i.ctypes.data_as(_fp)))
        return d.T.copy(), i.T.copy()

    def input_image(self, which, stream=0):
        shape, dt = {IN_DEPTH_MM: ((self.rows, self.cols), np.uint16), IN_DEPTH_FILTERED_MM: ((self.rows, self.cols), np.uint16),
                     IN_DEPTH_METRIC: ((self.rows, self.cols), np.float32), IN_COLOR: ((self.rows, self.cols, 3), np.uint8)}[which]
        out = np.zeros(shape, dtype=dt)
        self.api.check(self.api.get_input_image(self.h, stream, which, out.ctypes.data_as(C.c_void_p)))
        return out

    # -- frame-to-model prediction ----------------------------------------
    def default_model_params(self):
        p = SfModelParams()
        self.api.check(self.api.default_model_params(self.h, C.byref(p)))
        return p

    def predict_from_model(self, stream, surfels, pose, params=None):
        """surfels: (count, 12) float32 in the reference's vertex layout; pose: 4x4 (row, col) camera pose"""
        s = np.ascontiguousarray(surfels, dtype=np.float32).reshape(-1, 12)
        T = np.ascontiguousarray(np.asarray(pose, np.float32).T)  # column-major storage
        p = params if params is not None else self.default_model_params()
        self.api.check(self.api.predict_from_model(self.h, stream, s.ctypes.data_as(_fp), s.shape[0], T.ctypes.data_as(_fp), C.byref(p)))

    def init_model_from_frame(self, stream, pose, params=None, time=1):
        """GlobalModel::initialise on the frame the stream holds -> (count, 12) surfels in the reference's vertex layout"""
        T = np.ascontiguousarray(np.asarray(pose, np.float32).T)
        p = params if params is not None else self.default_model_params()
        out = np.zeros((self.rows * self.cols, 12), np.float32)
        n = C.c_int32()
        self.api.check(self.api.init_model_from_frame(self.h, stream, T.ctypes.data_as(_fp), C.byref(p), time, out.ctypes.data_as(_fp), C.byref(n)))
        return out[: n.value].copy()

    def prediction_dense(self):
        """Reconstruction::denseEnough of the last prediction (what checkIfDenseEnough reports one frame later)"""
        d = C.c_int32()
        self.api.check(self.api.get_prediction_dense(self.h, C.byref(d)))
        return bool(d.value)

    def prediction_dense_stream(self, stream):
        d = C.c_int32()
        self.api.check(self.api.get_prediction_dense_stream(self.h, stream, C.byref(d)))
        return bool(d.value)

    def prediction(self, stream=0):
        """(depthPrediction, intensityPrediction) as (rows, cols) arrays"""
        d = np.zeros((self.cols, self.rows), dtype=np.float32)
        i = np.zeros((self.cols, self.rows), dtype=np.float32)
        self.api.check(self.api.get_prediction(self.h, stream, d.ctypes.data_as(_fp), i.ctypes.data_as(_fp)))
        return d.T.copy(), i.T.copy()

    def advance_sequences_device(self, pool_depth_ptr, pool_intensity_ptr, frame_index, pool_frames):
        """prediction := current; current := pool frame frame_index[b] (device pools [pool_frames][cols][rows]; host index array)"""
        idx = np.ascontiguousarray(frame_index, dtype=np.int32)
        assert idx.shape == (self.batch_size,)
        self.api.check(self.api.advance_sequences_device(self.h, C.c_void_p(pool_depth_ptr), C.c_void_p(pool_intensity_ptr), idx.ctypes.data_as(_ip),
                                                         int(pool_frames)))

    def _traj(self, n_frames, want):
        return np.zeros((n_frames, self.batch_size, 16), np.float32) if want else None

    @staticmethod
    def _traj_out(T):
        # [frame][stream] 4x4 (row, column) matrices from the column-major storage
        return None if T is None else np.ascontiguousarray(T.reshape(T.shape[0], T.shape[1], 4, 4).transpose(0, 1, 3, 2))

    def process_frames(self, im_count0, n_frames, trajectory=False):
        """n_frames x process_frame in one launch (sf_process_frames); trajectory=True returns T of every frame and stream"""
        T = self._traj(n_frames, trajectory)
        self.api.check(self.api.process_frames(self.h, im_count0, n_frames, T.ctypes.data_as(_fp) if trajectory else None))
        return self._traj_out(T)

    def process_sequence_frames_device(self, pool_depth_ptr, pool_intensity_ptr, frame_index, pool_frames, im_count0, trajectory=False):
        """frame_index: [n_frames][batch] pool frames; per frame and stream the advance step, then process_frame; one launch"""
        idx = np.ascontiguousarray(frame_index, dtype=np.int32)
        assert idx.ndim == 2 and idx.shape[1] == self.batch_size
        T = self._traj(idx.shape[0], trajectory)
        self.api.check(self.api.process_sequence_frames_device(self.h, C.c_void_p(pool_depth_ptr), C.c_void_p(pool_intensity_ptr), idx.ctypes.data_as(_ip),
                                                               int(pool_frames), im_count0, idx.shape[0], T.ctypes.data_as(_fp) if trajectory else None))
        return self._traj_out(T)

    def current_to_prediction(self):
        self.api.check(self.api.current_to_prediction(self.h))

    def set_segm_state(self, stream, labels0=None, b_segm=None, cluster_res=None):
        lab = None if labels0 is None else np.ascontiguousarray(np.asarray(labels0, dtype=np.int32).T)
        bb = None if b_segm is None else np.ascontiguousarray(b_segm, dtype=np.float32)
        cr = None if cluster_res is None else np.ascontiguousarray(cluster_res, dtype=np.float32)
        self.api.check(self.api.set_segm_state(
            self.h, stream,
            None if lab is None else lab.ctypes.data_as(_ip),
            None if bb is None else bb.ctypes.data_as(_fp),
            None if cr is None else cr.ctypes.data_as(_fp)))

    def set_twist_old(self, stream, twist):
        t = np.ascontiguousarray(twist, dtype=np.float32)
        self.api.check(self.api.set_twist_old(self.h, stream, t.ctypes.data_as(_fp)))

    # -- the reference methods ------------------------------------------------------------------
    def build_pyramid(self, old_im):
        self.api.check(self.api.build_pyramid(self.h, int(bool(old_im))))

    def kmeans(self):
        self.api.check(self.api.kmeans(self.h))

    def run_solver(self, create_image_pyr=True):
        self.api.check(self.api.run_solver(self.h, int(bool(create_image_pyr))))

    def push_history(self, im_count):
        self.api.check(self.api.push_history(self.h, im_count))

    def residuals_vs_history(self, index):
        self.api.check(self.api.residuals_vs_history(self.h, index))

    def build_segm_image(self):
        self.api.check(self.api.build_segm_image(self.h))

    def process_frame(self, im_count):
        self.api.check(self.api.process_frame(self.h, im_count))

    def synchronize(self):
        self.api.check(self.api.synchronize(self.h))

    # -- outputs --------------------------------------------------------------------------------
    def _vec(self, fn, stream, n):
        out = np.zeros(n, dtype=np.float32)
        self.api.check(fn(self.h, stream, out.ctypes.data_as(_fp)))
        return out

    def T(self, stream=0):
        return self._vec(self.api.get_T, stream, 16).reshape(4, 4).T.copy()  # column-major -> [r, c]

    def twist(self, stream=0):
        return self._vec(self.api.get_twist, stream, 6)

    def twist_old(self, stream=0):
        return self._vec(self.api.get_twist_old, stream, 6)

    def b(self, stream=0):
        return self._vec(self.api.get_b, stream, NUM_CLUSTERS)

    def cluster_residuals(self, stream=0):
        return self._vec(self.api.get_cluster_residuals, stream, NUM_CLUSTERS)

    def kmeans_centres(self, stream=0):
        return self._vec(self.api.get_kmeans, stream, 3 * NUM_CLUSTERS).reshape(NUM_CLUSTERS, 3).T.copy()  # (3, 24)

    def b_image(self, stream=0):
        out = np.zeros((self.cols, self.rows), dtype=np.float32)
        self.api.check(self.api.get_b_image(self.h, stream, out.ctypes.data_as(_fp)))
        return out.T.copy()

    def level_shape(self, level):
        return self.api.level_rows(self.h, level), self.api.level_cols(self.h, level)

    def labels(self, level, stream=0):
        r, c = self.level_shape(level)
        out = np.zeros((c, r), dtype=np.int32)
        self.api.check(self.api.get_labels(self.h, stream, level, out.ctypes.data_as(_ip)))
        return out.T.copy()

    def connectivity(self, stream=0):
        out = np.zeros((NUM_CLUSTERS, NUM_CLUSTERS), dtype=np.uint8)
        self.api.check(self.api.get_connectivity(self.h, stream, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out.astype(bool)

    def plane(self, pset, channel, level, stream=0):
        r, c = self.level_shape(level)
        out = np.zeros((c, r), dtype=np.float32)
        self.api.check(self.api.get_plane(self.h, stream, pset, channel, level, out.ctypes.data_as(_fp)))
        return out.T.copy()

    def lin_plane(self, which, stream=0):
        r, c = C.c_int32(), C.c_int32()
        self.api.check(self.api.get_lin_plane(self.h, stream, which, None, C.byref(r), C.byref(c)))
        out = np.zeros((c.value, r.value), dtype=np.float32)
        self.api.check(self.api.get_lin_plane(self.h, stream, which, out.ctypes.data_as(_fp), C.byref(r), C.byref(c)))
        return out.T.copy()

    def jacobian_rows(self, stream=0):
        """(A (2N, 6), B (2N,)) of the last executed outer iteration, validPixels order (needs debug_planes)"""
        n = C.c_int32()
        self.api.check(self.api.get_jacobian_rows(self.h, stream, None, None, C.byref(n)))
        A = np.zeros((max(n.value, 1), 6), np.float32)
        B = np.zeros(max(n.value, 1), np.float32)
        self.api.check(self.api.get_jacobian_rows(self.h, stream, A.ctypes.data_as(_fp), B.ctypes.data_as(_fp), C.byref(n)))
        return A[: n.value].copy(), B[: n.value].copy()

    def stats(self, stream=0):
        st = SfFrameStats()
        self.api.check(self.api.get_stats(self.h, stream, C.byref(st)))
        return st

    def batch_results(self):
        B = self.batch_size
        T = np.zeros((B, 16), dtype=np.float32)
        n_irls = np.zeros(B, dtype=np.int32)
        n_outer = np.zeros(B, dtype=np.int32)
        pix = np.zeros(B, dtype=np.int64)
        self.api.check(
            self.api.get_batch_results(
                self.h, T.ctypes.data_as(_fp), n_irls.ctypes.data_as(_ip), n_outer.ctypes.data_as(_ip),
                pix.ctypes.data_as(C.POINTER(C.c_int64)),
            )
        )
        return T.reshape(B, 4, 4).transpose(0, 2, 1).copy(), n_irls, n_outer, pix

    def counters(self):
        """(frames, n_irls, n_outer, pixel_iters) totals since creation, over all streams"""
        v = [C.c_int64() for _ in range(4)]
        self.api.check(self.api.get_counters(self.h, *[C.byref(x) for x in v]))
        return tuple(int(x.value) for x in v)

    STAGES = ["pyr_old", "pyr_new", "kmeans", "warp", "linearise", "irls_setup", "pass1", "solve6", "pass2",
              "b_solve", "filter", "residuals", "segm_hist", "total", "km_init", "km_sort", "km_assign", "km_partition",
              "km_sum", "km_label0", "km_conn_pyr"]

    def stage_profile(self):
        """dict stage -> seconds (lane-0 wall clock summed over streams since creation)"""
        t = (C.c_int64 * 32)()
        self.api.check(self.api.get_stage_profile(self.h, t))
        return {n: t[i] * 1e-8 for i, n in enumerate(self.STAGES)}

    def splat_replays(self):
        """warp tiles replayed for targets outside their accumulation window (slot 24 of the stage profile: a counter)"""
        t = (C.c_int64 * 32)()
        self.api.check(self.api.get_stage_profile(self.h, t))
        return int(t[24])

    def ordered_fallbacks(self):
        """levels whose ordered tile splat gave up and took the per-cell source lists (slot 25 of the stage profile: a counter)"""
        t = (C.c_int64 * 32)()
        self.api.check(self.api.get_stage_profile(self.h, t))
        return int(t[25])

    def shader_clock_counters(self):
        """(shader cycles, 100 MHz ticks) the workgroups have spent inside stream-frames since creation (slots 26 and 13 of
        the stage profile): the difference of two readings gives the clock the frames in between ran at,
        100 * cycles / ticks MHz -- what the package's power management granted, measured where the work ran"""
        t = (C.c_int64 * 32)()
        self.api.check(self.api.get_stage_profile(self.h, t))
        return int(t[26]), int(t[13])

    @staticmethod
    def shader_clock_mhz(before, after):
        return 100.0 * (after[0] - before[0]) / max(1, after[1] - before[1])

    def microbench_pass(self, which, variant, reps):
        ms = C.c_float()
        self.api.check(self.api.microbench_pass(self.h, which, variant, reps, C.byref(ms)))
        return ms.value

    def timed_process_frames(self, im_count, calls):
        ms = C.c_float()
        self.api.check(self.api.timed_process_frames(self.h, im_count, calls, C.byref(ms)))
        return ms.value

    def clear_sync_timeout(self):
        self.api.check(self.api.clear_sync_timeout(self.h))

    def debug_stall_rank(self, rank, stall_ms=0.0, spin_limit=0):
        self.api.check(self.api.debug_stall_rank(self.h, rank, stall_ms, spin_limit))

    def microbench_copy(self, nbytes, reps):
        """GB/s of `reps` device copies of nbytes bytes (2 x nbytes moved each): the measured streaming ceiling of this box"""
        ms = C.c_float()
        self.api.check(self.api.microbench_copy(self.h, int(nbytes), int(reps), C.byref(ms)))
        return 2.0 * nbytes * reps / (ms.value * 1e-3) / 1e9

    def last_solver_kernel_ms(self):
        ms = C.c_float()
        self.api.check(self.api.last_solver_kernel_ms(self.h, C.byref(ms)))
        return ms.value


class SurfelMap:
    """numpy wrapper over one sf_map (GlobalModel + currPose / tick of Reconstruction), SURVEY.md §8(f) rank 4"""

    def __init__(self, solver, capacity=0):
        self.solver, self.api = solver, solver.api
        self.m = C.c_void_p()
        self.api.check(self.api.map_create(solver.h, capacity, C.byref(self.m)))

    def close(self):
        if self.m:
            self.api.map_destroy(self.m)
            self.m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fuse_frame(self, stream, in_pose, weight_multiplier=1.0, params=None):
        """Reconstruction::fuseFrame for the frame `stream` holds; in_pose: 4x4 (row, col) T_odometry or None (first call)"""
        p = params if params is not None else self.solver.default_model_params()
        T = None if in_pose is None else np.ascontiguousarray(np.asarray(in_pose, np.float32).T)
        self.api.check(self.api.map_fuse_frame(self.solver.h, stream, self.m, None if T is None else T.ctypes.data_as(_fp),
                                               weight_multiplier, C.byref(p)))

    def predict(self, stream, params=None):
        p = params if params is not None else self.solver.default_model_params()
        self.api.check(self.api.map_predict(self.solver.h, stream, self.m, C.byref(p)))

    @staticmethod
    def _batch(streams, maps):
        n = len(maps)
        assert len(streams) == n
        return n, (C.c_int32 * n)(*streams), (C.c_void_p * n)(*[m.m.value for m in maps])

    @staticmethod
    def fuse_frames(solver, streams, maps, in_poses, weight_multiplier=1.0, params=None):
        """sf_map_fuse_frames: streams[q] into maps[q]; in_poses: list of 4x4 (row, col) or None (all maps at tick 1)"""
        p = params if params is not None else solver.default_model_params()
        n, s, m = SurfelMap._batch(streams, maps)
        T = None if in_poses is None else np.ascontiguousarray(np.stack([np.asarray(t, np.float32).T for t in in_poses]))
        solver.api.check(solver.api.map_fuse_frames(solver.h, n, s, m, None if T is None else T.ctypes.data_as(_fp), weight_multiplier, C.byref(p)))

    @staticmethod
    def predict_frames(solver, streams, maps, params=None):
        p = params if params is not None else solver.default_model_params()
        n, s, m = SurfelMap._batch(streams, maps)
        solver.api.check(solver.api.map_predict_frames(solver.h, n, s, m, C.byref(p)))

    def info(self):
        """dict(count, tick, pose (4x4 row, col), stats)"""
        n, t = C.c_int32(), C.c_int32()
        pose = np.zeros(16, np.float32)
        stats = (C.c_int32 * 4)()
        self.api.check(self.api.map_info(self.m, C.byref(n), C.byref(t), pose.ctypes.data_as(_fp), stats))
        return dict(count=n.value, tick=t.value, pose=pose.reshape(4, 4).T.copy(), stats=[int(x) for x in stats])

    def download(self):
        n = self.info()["count"]
        out = np.zeros((max(n, 1), 12), np.float32)
        self.api.check(self.api.map_download(self.m, out.ctypes.data_as(_fp), n))
        return out[:n].copy()

    def upload(self, surfels, pose, tick):
        s = np.ascontiguousarray(surfels, dtype=np.float32).reshape(-1, 12)
        T = np.ascontiguousarray(np.asarray(pose, np.float32).T)
        self.api.check(self.api.map_upload(self.m, s.ctypes.data_as(_fp), s.shape[0], T.ctypes.data_as(_fp), tick))

    def index_map(self):
        out = np.zeros((self.solver.rows * 4, self.solver.cols * 4), np.uint32)
        self.api.check(self.api.map_get_index_map(self.m, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out
