"""gpismap_amd -- MI355X-native GPisMap hot path (ObsGP + OnGPIS on hand-written HIP kernels).

Thin ctypes binding of the C-ABI in include/gpismap_amd.h.  The classes mirror the command set of
the reference's mex gateways (mex/mexGPisMap3.cpp: 'setCamera' / 'update' / 'test' /
'getAllPoints' / 'reset').  There is NO CPU fallback: if libgpismap_amd.so is missing, or no HIP
device is present, the compute entry points raise.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPISMAP_AMD_LIB", os.path.join(_HERE, "libgpismap_amd.so"))
_lib = None

fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int)
dp = C.POINTER(C.c_double)


class GpisError(RuntimeError):
    pass


class gpis_cam(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("width", C.c_int), ("height", C.c_int)]


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


def lib():
    """Load the native library (built in-tree by __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GpisError("native library %s not built; run __graft_entry__.build()" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.gpis_device_count.restype = C.c_int
    L.gpis_pool_cache_trim.restype = C.c_ulonglong
    L.gpis_pool_cache_trim.argtypes = []
    L.gpis_version.restype = C.c_char_p
    L.gpis_set_device.argtypes = [C.c_int]
    L.gpis3_device.argtypes = [vp]
    L.gpis3_set_shard.argtypes = [vp, C.c_int, C.c_int]
    L.gpis3_shard_info.argtypes = [vp, ip, C.c_int]
    L.gpis3_shard_bytes.argtypes = [vp, C.c_int]; L.gpis3_shard_bytes.restype = C.c_longlong
    L.gpis3_shard_pack.argtypes = [vp, vp, vp]
    L.gpis3_shard_unpack.argtypes = [vp, C.c_int, vp, vp]
    L.gpis3_shard_finish.argtypes = [vp]
    if hasattr(L, "gpis3_apply_frame"):
        L.gpis3_set_frame_export.argtypes = [vp, C.c_int]
        L.gpis3_frame_record.argtypes = [vp, vp, C.c_longlong]
        L.gpis3_frame_record.restype = C.c_longlong
        L.gpis3_train_deferred.argtypes = [vp]
        L.gpis3_apply_frame.argtypes = [vp, vp, C.c_longlong]
    L.gpis_ongpis_packed_bytes.argtypes = [vp, ip, C.c_int]; L.gpis_ongpis_packed_bytes.restype = C.c_longlong
    L.gpis_ongpis_pack.argtypes = [vp, ip, C.c_int, vp, C.c_longlong, vp]
    L.gpis_ongpis_unpack.argtypes = [vp, vp, C.c_int, C.c_longlong, ip, vp]
    L.gpis2_device.argtypes = [vp]
    L.gpis3_create.restype = vp
    L.gpis3_create.argtypes = [C.POINTER(gpis_cam)]
    L.gpis3_destroy.argtypes = [vp]
    L.gpis3_reset.argtypes = [vp]
    L.gpis3_set_camera.argtypes = [vp, C.POINTER(gpis_cam)]
    L.gpis3_update.argtypes = [vp, fp, C.c_int, fp]
    L.gpis3_test.argtypes = [vp, fp, C.c_int, C.c_int, fp]
    L.gpis3_test_device.argtypes = [vp, vp, C.c_int, vp, vp]
    L.gpis3_num_points.argtypes = [vp]
    L.gpis3_save.argtypes = [vp, C.c_char_p]
    L.gpis3_load.argtypes = [vp, C.c_char_p]
    L.gpis3_get_points.argtypes = [vp, fp, C.c_int]
    L.gpis3_get_nodes.argtypes = [vp, fp, C.c_int]
    L.gpis3_stats.argtypes = [vp, dp, C.c_int]
    L.gpis3_set_profile.argtypes = [vp, C.c_int]
    L.gpis3_sync.argtypes = [vp]
    L.gpis3_set_pipeline.argtypes = [vp, C.c_int]
    L.gpis3_set_host_gather.argtypes = [vp, C.c_int]
    L.gpis3_set_keep_factors.argtypes = [vp, C.c_int]
    L.gpis3_set_shard_factors.argtypes = [vp, C.c_int]
    L.gpis2_create.restype = vp
    L.gpis2_destroy.argtypes = [vp]
    L.gpis2_reset.argtypes = [vp]
    L.gpis2_update.argtypes = [vp, fp, fp, C.c_int, fp]
    L.gpis2_test.argtypes = [vp, fp, C.c_int, C.c_int, fp]
    L.gpis2_test_device.argtypes = [vp, vp, C.c_int, vp, vp]
    L.gpis2_get_nodes.argtypes = [vp, fp, C.c_int]
    L.gpis2_stats.argtypes = [vp, dp, C.c_int]
    if hasattr(L, "gpis2_sync"):
        L.gpis2_sync.argtypes = [vp]
        L.gpis2_set_pipeline.argtypes = [vp, C.c_int]
    L.gpis_obsgp_create.restype = vp
    L.gpis_obsgp_destroy.argtypes = [vp]
    L.gpis_obsgp_train2d.argtypes = [vp, fp, fp, C.c_int, C.c_int]
    L.gpis_obsgp_train1d.argtypes = [vp, fp, fp, C.c_int]
    L.gpis_obsgp_query.argtypes = [vp, fp, C.c_int, fp, fp]
    L.gpis_obsgp_num_groups.argtypes = [vp]
    L.gpis_obsgp_get_group.argtypes = [vp, C.c_int, ip, fp, fp, fp]
    L.gpis_ongpis_create.restype = vp
    L.gpis_ongpis_create.argtypes = [C.c_int, C.c_float]
    L.gpis_ongpis_destroy.argtypes = [vp]
    L.gpis3_create_multi.restype = vp
    L.gpis3_create_multi.argtypes = [C.c_void_p, ip, C.c_int]
    L.gpis3_num_devices.argtypes = [vp]
    L.gpis_ongpis_train.argtypes = [vp, fp, C.c_int, ip, ip, C.c_int, ip]
    L.gpis_ongpis_model_dims.argtypes = [vp, C.c_int, ip]
    L.gpis_ongpis_get_model.argtypes = [vp, C.c_int, fp, fp, ip]
    L.gpis_ongpis_eval.argtypes = [vp, fp, C.c_int, ip, ip, C.c_int, fp]
    L.gpis_ongpis_last_ms.argtypes = [vp, fp, fp]
    L.gpis_ongpis_set_exp_table.argtypes = [vp, C.c_int]
    L.gpis_ongpis_kernel_matrix.argtypes = [vp, fp, ip, fp, fp, C.c_int, fp]
    L.gpis_ongpis_set_debug.argtypes = [vp, C.c_int, C.c_int]
    if hasattr(L, "gpis_ongpis_set_cu_reserve"):
        L.gpis_ongpis_set_cu_reserve.argtypes = [vp, C.c_int]
    if hasattr(L, "gpis_selftest_ranged_arith"):       # (A/B runs load older builds of the library through GPISMAP_AMD_LIB)
        L.gpis_selftest_ranged_arith.argtypes = [C.c_ulonglong, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
        L.gpis_selftest_ranged_arith.restype = C.c_int
    L.gpis_ongpis_set_keep_factor.argtypes = [vp, C.c_int]
    L.gpis_ongpis_set_fused.argtypes = [vp, C.c_int]
    L.gpis_ongpis_set_lazy_inverse.argtypes = [vp, C.c_int]
    L.gpis3_prepare_test.argtypes = [vp]
    L.gpis3_set_lazy_inverse.argtypes = [vp, C.c_int]
    _lib = L
    return L


def selftest_ranged_arith(seed=1, blocks=1024, per_thread=64, mode=0):
    """gpis_selftest_ranged_arith: (square-root mismatches, division mismatches) of blocks * 256 * per_thread operand pairs run
    through the factorisation kernels' range-restricted sqrt / division and through the compiler's IEEE ones on the device."""
    m = (C.c_ulonglong * 2)(0, 0)
    _check(lib().gpis_selftest_ranged_arith(int(seed), int(blocks), int(per_thread), int(mode), m), "gpis_selftest_ranged_arith")
    return int(m[0]), int(m[1])


def pool_cache_trim():
    """Hand the device-pool chunks the library caches across maps back to the driver (gpis_pool_cache_trim); bytes released."""
    return int(lib().gpis_pool_cache_trim()) if (_lib is not None or os.path.exists(LIB_PATH)) else 0


def _trim_at_exit():
    # only if the library was ever loaded: the cache holds memory of DESTROYED pools, nothing a live map uses
    if _lib is not None:
        try:
            _lib.gpis_pool_cache_trim()
        except Exception:
            pass


import atexit as _atexit  # noqa: E402
_atexit.register(_trim_at_exit)


def device_count():
    return lib().gpis_device_count()


def set_device(device):
    """Select the HIP device for every object created afterwards (one process per GPU: LOCAL_RANK)."""
    _check(lib().gpis_set_device(int(device)), "gpis_set_device(%d)" % int(device))


def get_device():
    return lib().gpis_get_device()


def _check(rc, what):
    if rc != 0:
        raise GpisError("%s failed with status %d" % (what, rc))


def _cam(cam6):
    c = np.asarray(cam6, dtype=np.float64)
    return gpis_cam(float(c[0]), float(c[1]), float(c[2]), float(c[3]), int(c[4]), int(c[5]))


class GPisMap3:
    """Mirror of the reference's mexGPisMap3 command set on the HIP path."""

    STAT_KEYS = ("obsgp_groups", "obsgp_queries", "clusters_trained", "late_reevals", "clusters",
                 "last_test_evals", "last_test_k4_ms", "device_bytes", "last_test_flops", "last_test_k4_launches",
                 "last_train_ms", "model_bytes", "upd_preproc_ms", "upd_obsgp_train_ms", "upd_reeval_ms", "upd_eval_ms",
                 "upd_gps_ms", "last_train_flops", "last_train_bytes", "last_train_jobs", "last_train_maxK",
                 "last_inverse_ms", "last_inverse_jobs", "exchange_bytes", "pipelined", "train_cu_reserve", "host_replays", "deferred_inverses")

    def __init__(self, cam6=None, devices=None):
        """devices: list of HIP device ids for ONE map over several devices (gpis3_create_multi; a device may repeat:
        logical shards on one GPU); None = the current device (or GPIS_DEVICES from the environment)."""
        self.L = lib()
        if self.L.gpis_device_count() < 1:
            raise GpisError("no HIP device: gpismap_amd has no CPU fallback")
        cam = C.byref(_cam(cam6)) if cam6 is not None else None
        if devices is None:
            self.h = C.c_void_p(self.L.gpis3_create(cam))
        else:
            d = np.ascontiguousarray(devices, dtype=np.int32)
            self.h = C.c_void_p(self.L.gpis3_create_multi(cam, _p(d, C.c_int), d.size))
        if not self.h:
            raise GpisError("gpis3_create failed")

    def num_devices(self):
        return int(self.L.gpis3_num_devices(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.L.gpis3_destroy(self.h)
            self.h = None

    __del__ = close

    def reset(self):
        _check(self.L.gpis3_reset(self.h), "gpis3_reset")

    def set_camera(self, cam6):
        _check(self.L.gpis3_set_camera(self.h, C.byref(_cam(cam6))), "gpis3_set_camera")

    def update(self, depth, pose):
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        pose = np.ascontiguousarray(pose, dtype=np.float32)
        if pose.size != 12:
            raise GpisError("pose must have 12 elements")
        _check(self.L.gpis3_update(self.h, _p(depth), depth.size, _p(pose)), "gpis3_update")

    def test(self, x, res=None):
        """x: [N,3] float32.  Returns res [N,8] (zero pre-filled like the mex gateway) or None
        where the reference's test() returns false."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if res is None:
            res = np.zeros((x.shape[0], 8), dtype=np.float32)
        rc = self.L.gpis3_test(self.h, _p(x), 3, x.shape[0], _p(res))
        if rc == -1:
            return None
        _check(rc, "gpis3_test")
        return res

    def test_device(self, d_x_ptr, n, d_res_ptr, stream=0):
        _check(self.L.gpis3_test_device(self.h, C.c_void_p(d_x_ptr), n, C.c_void_p(d_res_ptr), C.c_void_p(stream)),
               "gpis3_test_device")

    def device(self):
        return self.L.gpis3_device(self.h)

    # ---- sharded training (include/gpismap_amd.h, gpis3_set_shard ...) ----
    def set_shard(self, rank, world):
        _check(self.L.gpis3_set_shard(self.h, int(rank), int(world)), "gpis3_set_shard")
        self._world = int(world)

    def shard_info(self):
        w = getattr(self, "_world", 1)
        out = np.zeros(2 + w, dtype=np.int32)
        _check(self.L.gpis3_shard_info(self.h, _p(out, C.c_int), 2 + w), "gpis3_shard_info")
        return int(out[0]), int(out[1]), [int(v) for v in out[2:]]

    def shard_bytes(self, owner):
        """Bytes of rank `owner`'s packed records of the last update (records back to back at their own sizes; every rank
        gives the same answer for every owner)."""
        b = int(self.L.gpis3_shard_bytes(self.h, int(owner)))
        if b < 0:
            raise GpisError("gpis3_shard_bytes failed (%d)" % b)
        return b

    def shard_pack(self, d_buf_ptr, stream=0):
        _check(self.L.gpis3_shard_pack(self.h, C.c_void_p(d_buf_ptr), C.c_void_p(stream)), "gpis3_shard_pack")

    def shard_unpack(self, owner, d_buf_ptr, stream=0):
        _check(self.L.gpis3_shard_unpack(self.h, int(owner), C.c_void_p(d_buf_ptr), C.c_void_p(stream)), "gpis3_shard_unpack")

    def shard_finish(self):
        _check(self.L.gpis3_shard_finish(self.h), "gpis3_shard_finish")

    # ---- one process per GPU, host logic once (gpis3_set_frame_export ...; gpismap_amd.sharding.update_lead_worker) ----
    def set_frame_export(self, on=True):
        _check(self.L.gpis3_set_frame_export(self.h, 1 if on else 0), "gpis3_set_frame_export")

    def frame_record(self):
        """The record of the last update() on the lead (numpy uint8)."""
        n = int(self.L.gpis3_frame_record(self.h, None, 0))
        if n < 0:
            raise GpisError("gpis3_frame_record failed (%d)" % n)
        buf = np.empty(max(n, 1), dtype=np.uint8)
        m = int(self.L.gpis3_frame_record(self.h, buf.ctypes.data_as(C.c_void_p), n))
        if m != n:
            raise GpisError("gpis3_frame_record failed (%d)" % m)
        return buf[:n]

    def train_deferred(self):
        _check(self.L.gpis3_train_deferred(self.h), "gpis3_train_deferred")

    def apply_frame(self, record):
        record = np.ascontiguousarray(record, dtype=np.uint8)
        _check(self.L.gpis3_apply_frame(self.h, record.ctypes.data_as(C.c_void_p), int(record.size)), "gpis3_apply_frame")

    def num_points(self):
        return self.L.gpis3_num_points(self.h)

    def get_all_points(self):
        n = self.L.gpis3_get_points(self.h, None, 0)
        out = np.zeros((n, 3), dtype=np.float32)
        if n:
            self.L.gpis3_get_points(self.h, _p(out), n)
        return out

    def nodes(self):
        n = self.L.gpis3_get_nodes(self.h, None, 0)
        out = np.zeros((n, 9), dtype=np.float32)
        if n:
            self.L.gpis3_get_nodes(self.h, _p(out), n)
        return out

    def stats(self):
        a = (C.c_double * 28)()
        _check(self.L.gpis3_stats(self.h, a, 28), "gpis3_stats")
        return dict(zip(self.STAT_KEYS, list(a)))

    def save(self, path):
        """Map checkpoint: spatial index, surface points and the packed prediction records of the trained models (gpis3_save)."""
        _check(self.L.gpis3_save(self.h, os.fsencode(path)), "gpis3_save")

    def load(self, path):
        """Replace the map's state with a checkpoint's; models are restored verbatim, nothing is retrained (gpis3_load)."""
        _check(self.L.gpis3_load(self.h, os.fsencode(path)), "gpis3_load")

    def set_profile(self, on=True):
        _check(self.L.gpis3_set_profile(self.h, int(on)), "gpis3_set_profile")

    def prepare_test(self):
        """Join a pipelined training and compute the inverses the last updates left to the first test()."""
        _check(self.L.gpis3_prepare_test(self.h), "gpis3_prepare_test")

    def set_lazy_inverse(self, on=True):
        _check(self.L.gpis3_set_lazy_inverse(self.h, int(on)), "gpis3_set_lazy_inverse")

    def sync(self):
        """Join the training the last update() left in flight (pipelined update, include/gpismap_amd.h)."""
        _check(self.L.gpis3_sync(self.h), "gpis3_sync")

    def set_pipeline(self, on=True):
        _check(self.L.gpis3_set_pipeline(self.h, int(on)), "gpis3_set_pipeline")

    def set_host_gather(self, on=True):
        _check(self.L.gpis3_set_host_gather(self.h, int(on)), "gpis3_set_host_gather")

    def set_shard_factors(self, mode=-1):
        """Records of a sharded update: 1 factor records (receivers invert lazily), 0 prediction records, -1 follow the inverse mode."""
        _check(self.L.gpis3_set_shard_factors(self.h, int(mode)), "gpis3_set_shard_factors")

    def set_keep_factors(self, on=True):
        """Cross-check switch: keep the training side (factor, re-tiled factor) of every model after its inverse exists."""
        _check(self.L.gpis3_set_keep_factors(self.h, int(on)), "gpis3_set_keep_factors")


class GPisMap:
    """Mirror of the reference's mexGPisMap command set ('update', 'test', 'reset') on the HIP path."""

    def __init__(self):
        self.L = lib()
        if self.L.gpis_device_count() < 1:
            raise GpisError("no HIP device: gpismap_amd has no CPU fallback")
        self.h = C.c_void_p(self.L.gpis2_create())
        if not self.h:
            raise GpisError("gpis2_create failed")

    def close(self):
        if getattr(self, "h", None):
            self.L.gpis2_destroy(self.h)
            self.h = None

    __del__ = close

    def reset(self):
        _check(self.L.gpis2_reset(self.h), "gpis2_reset")

    def update(self, thetas, ranges, pose6):
        thetas = np.ascontiguousarray(thetas, dtype=np.float32)
        ranges = np.ascontiguousarray(ranges, dtype=np.float32)
        pose6 = np.ascontiguousarray(pose6, dtype=np.float32)
        if pose6.size != 6 or thetas.size != ranges.size:
            raise GpisError("bad 2-D update arguments")
        _check(self.L.gpis2_update(self.h, _p(thetas), _p(ranges), ranges.size, _p(pose6)), "gpis2_update")

    def test(self, x, res=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        if res is None:
            res = np.zeros((x.shape[0], 6), dtype=np.float32)
        rc = self.L.gpis2_test(self.h, _p(x), 2, x.shape[0], _p(res))
        if rc == -1:
            return None
        _check(rc, "gpis2_test")
        return res

    def nodes(self):
        n = self.L.gpis2_get_nodes(self.h, None, 0)
        out = np.zeros((n, 7), dtype=np.float32)
        if n:
            self.L.gpis2_get_nodes(self.h, _p(out), n)
        return out

    def stats(self):
        a = (C.c_double * 12)()
        _check(self.L.gpis2_stats(self.h, a, 12), "gpis2_stats")
        return dict(zip(GPisMap3.STAT_KEYS, list(a)))

    def sync(self):
        """Join the training the last update() left in flight (pipelined mode); raises when it failed."""
        _check(self.L.gpis2_sync(self.h), "gpis2_sync")

    def set_pipeline(self, on=True):
        _check(self.L.gpis2_set_pipeline(self.h, 1 if on else 0), "gpis2_set_pipeline")


class ObsGP:
    """Kernel-level K1/K2: device-resident observation GP."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.gpis_obsgp_create())
        if not self.h:
            raise GpisError("gpis_obsgp_create failed (no HIP device?)")

    def close(self):
        if getattr(self, "h", None):
            self.L.gpis_obsgp_destroy(self.h)
            self.h = None

    __del__ = close

    def train2d(self, vu, f, ni, nj):
        vu = np.ascontiguousarray(vu, dtype=np.float32)
        f = np.ascontiguousarray(f, dtype=np.float32)
        _check(self.L.gpis_obsgp_train2d(self.h, _p(vu), _p(f), ni, nj), "gpis_obsgp_train2d")

    def train1d(self, theta, f):
        theta = np.ascontiguousarray(theta, dtype=np.float32)
        f = np.ascontiguousarray(f, dtype=np.float32)
        _check(self.L.gpis_obsgp_train1d(self.h, _p(theta), _p(f), theta.size), "gpis_obsgp_train1d")

    def query(self, q, val0=0.0):
        q = np.ascontiguousarray(q, dtype=np.float32)
        n = q.shape[0]
        val = np.full(n, val0, dtype=np.float32)
        var = np.zeros(n, dtype=np.float32)
        _check(self.L.gpis_obsgp_query(self.h, _p(q), n, _p(val), _p(var)), "gpis_obsgp_query")
        return val, var

    def num_groups(self):
        return self.L.gpis_obsgp_num_groups(self.h)

    def group(self, g):
        n = C.c_int(0)
        x = np.zeros((64, 2), dtype=np.float32)
        alpha = np.zeros(64, dtype=np.float32)
        L = np.zeros((64, 64), dtype=np.float32)  # column-major on the device: L[c, r]
        _check(self.L.gpis_obsgp_get_group(self.h, g, C.byref(n), _p(x), _p(alpha), _p(L)), "gpis_obsgp_get_group")
        return n.value, x, alpha, L.T.copy()


class OnGPIS:
    """Kernel-level K6/K3/K4: batched cluster training and prediction."""

    def __init__(self, dim, scale, keep_factor=False, fused=True):
        """keep_factor: models of at most 256 rows (trained on chip) also keep L / alpha / gidx for model();
        fused=False: every cluster takes the separate training kernels."""
        self.L = lib()
        self.dim = dim
        self.h = C.c_void_p(self.L.gpis_ongpis_create(dim, float(scale)))
        if not self.h:
            raise GpisError("gpis_ongpis_create failed (no HIP device?)")
        _check(self.L.gpis_ongpis_set_keep_factor(self.h, 1 if keep_factor else 0), "gpis_ongpis_set_keep_factor")
        _check(self.L.gpis_ongpis_set_fused(self.h, 1 if fused else 0), "gpis_ongpis_set_fused")

    def close(self):
        if getattr(self, "h", None):
            self.L.gpis_ongpis_destroy(self.h)
            self.h = None

    __del__ = close

    def train(self, points9, off, ids):
        """points9: [9, npts] SoA; off: [ncl+1]; ids: concatenated point ids.  Returns model slots."""
        points9 = np.ascontiguousarray(points9, dtype=np.float32)
        off = np.ascontiguousarray(off, dtype=np.int32)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        ncl = off.size - 1
        models = np.zeros(ncl, dtype=np.int32)
        _check(self.L.gpis_ongpis_train(self.h, _p(points9), points9.shape[1], _p(off, C.c_int), _p(ids, C.c_int), ncl,
                                        _p(models, C.c_int)), "gpis_ongpis_train")
        return models

    def model(self, slot):
        d = np.zeros(4, dtype=np.int32)
        _check(self.L.gpis_ongpis_model_dims(self.h, int(slot), _p(d, C.c_int)), "gpis_ongpis_model_dims")
        N, ng, K, ld = [int(v) for v in d]
        Lm = np.zeros((ld, ld), dtype=np.float32)
        alpha = np.zeros(K, dtype=np.float32)
        gidx = np.zeros(N, dtype=np.int32)
        _check(self.L.gpis_ongpis_get_model(self.h, int(slot), _p(Lm), _p(alpha), _p(gidx, C.c_int)), "gpis_ongpis_get_model")
        return dict(N=N, ng=ng, K=K, ld=ld, L=Lm.T.copy(), alpha=alpha, gidx=gidx)  # L[r, c]

    def eval(self, xq, job_q, job_model, return_status=False):
        """return_status=True: (status, out) instead of raising -- GPIS_ERR_STATE (-3, the kernels' error word) still delivers
        `out`, with the affected results NaN."""
        xq = np.ascontiguousarray(xq, dtype=np.float32)
        job_q = np.ascontiguousarray(job_q, dtype=np.int32)
        job_model = np.ascontiguousarray(job_model, dtype=np.int32)
        out = np.zeros((job_q.size, 8), dtype=np.float32)
        rc = self.L.gpis_ongpis_eval(self.h, _p(xq), xq.shape[0], _p(job_q, C.c_int), _p(job_model, C.c_int), job_q.size, _p(out))
        if return_status:
            return int(rc), out
        _check(rc, "gpis_ongpis_eval")
        return out

    def packed_bytes(self, models):
        models = np.ascontiguousarray(models, dtype=np.int32)
        return int(self.L.gpis_ongpis_packed_bytes(self.h, _p(models, C.c_int), models.size))

    def pack(self, models, d_buf_ptr, stride, stream=0):
        models = np.ascontiguousarray(models, dtype=np.int32)
        _check(self.L.gpis_ongpis_pack(self.h, _p(models, C.c_int), models.size, C.c_void_p(d_buf_ptr), int(stride), C.c_void_p(stream)),
               "gpis_ongpis_pack")

    def unpack(self, d_buf_ptr, n, stride, models=None, stream=0):
        """records -> predict-only models; returns their ids (new ones unless `models` names slots to reuse)."""
        ids = np.full(n, -1, dtype=np.int32) if models is None else np.ascontiguousarray(models, dtype=np.int32).copy()
        _check(self.L.gpis_ongpis_unpack(self.h, C.c_void_p(d_buf_ptr), int(n), int(stride), _p(ids, C.c_int), C.c_void_p(stream)),
               "gpis_ongpis_unpack")
        return ids

    def kernel_matrix(self, x, gidx, sigx, sigg):
        """Kernel matrix of the build kernel on caller-given arrays (no gather rule): returns K[r, c], lower triangle."""
        x = np.ascontiguousarray(x, dtype=np.float32); gidx = np.ascontiguousarray(gidx, dtype=np.int32)
        sigx = np.ascontiguousarray(sigx, dtype=np.float32); sigg = np.ascontiguousarray(sigg, dtype=np.float32)
        n = gidx.size
        K = n + self.dim * int((gidx >= 0).sum())
        out = np.zeros(K * K, dtype=np.float32)
        _check(self.L.gpis_ongpis_kernel_matrix(self.h, _p(x), _p(gidx, C.c_int), _p(sigx), _p(sigg), n, _p(out)), "gpis_ongpis_kernel_matrix")
        return out.reshape(K, K).T.copy()

    def set_debug(self, inject=0, wait_limit_ms=0):
        """Bound of the in-kernel waits (0 = default 2 s) and the test-only fault injection: inject bit 0 = the cooperative
        factorisation withholds a hand-over, bit 4 (16) = the first workgroup of every prediction launch withholds one ring signal."""
        _check(self.L.gpis_ongpis_set_debug(self.h, int(inject), int(wait_limit_ms)), "gpis_ongpis_set_debug")

    def set_cu_reserve(self, n):
        """CUs the training streams of this handle leave free (CU-masked streams, as the maps' pipelined update uses them)."""
        _check(self.L.gpis_ongpis_set_cu_reserve(self.h, int(n)), "gpis_ongpis_set_cu_reserve")

    def set_lazy_inverse(self, on=True):
        _check(self.L.gpis_ongpis_set_lazy_inverse(self.h, 1 if on else 0), "gpis_ongpis_set_lazy_inverse")

    def set_exp_table(self, on=True):
        _check(self.L.gpis_ongpis_set_exp_table(self.h, 1 if on else 0), "gpis_ongpis_set_exp_table")

    def last_ms(self):
        a, b = C.c_float(0), C.c_float(0)
        self.L.gpis_ongpis_last_ms(self.h, C.byref(a), C.byref(b))
        return a.value, b.value
