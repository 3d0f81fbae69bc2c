"""ctypes wrapper over the C oracle (see vnect_oracle.h).  TEST INFRASTRUCTURE."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VNECT_ORACLE_SO: another build of the same sources (the ASan + UBSan one, `make -C oracle asan`)
_SO = os.environ.get("VNECT_ORACLE_SO") or os.path.join(_HERE, "_build", "libvnect_oracle.so")
_lib = None

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_u8p = C.POINTER(C.c_uint8)


def build(force=False):
    """Compile the oracle with gcc (no GPU involved)."""
    srcs = [os.path.join(_HERE, f) for f in ("vnect_net.c", "vnect_post.c", "vnect_oracle.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.vo_net_create.restype = C.c_void_p
        L.vo_net_destroy.argtypes = [C.c_void_p]
        L.vo_net_set_weight.argtypes = [C.c_void_p, C.c_char_p, c_f32p, C.POINTER(C.c_int64), C.c_int]
        L.vo_net_options.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.vo_net_forward.argtypes = [C.c_void_p, c_f32p, C.c_int, c_f32p]
        L.vo_net_activation.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
        L.vo_net_activation.restype = c_f32p
        L.vo_net_error.argtypes = [C.c_void_p]
        L.vo_net_error.restype = C.c_char_p
        L.vo_sgemm.argtypes = [C.c_int] * 3 + [c_f32p, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int]
        L.vo_cvround.argtypes = [C.c_double]
        L.vo_resize_size.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.vo_resize_u8.argtypes = [c_u8p, C.c_int, C.c_int, C.c_int, C.c_double, c_u8p]
        L.vo_resize_f32.argtypes = [c_f32p, C.c_int, C.c_int, C.c_int, C.c_double, c_f32p]
        L.vo_resize_f64.argtypes = [c_f64p, C.c_int, C.c_int, C.c_int, C.c_double, c_f64p]
        L.vo_gen_input_batch.argtypes = [c_u8p, C.c_int, C.c_int, C.c_int64, c_f64p, C.c_int, c_f32p, c_f64p,
                                         C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.vo_merge_scales.argtypes = [c_f32p, c_f64p, C.c_int, c_f64p]
        L.vo_extract_2d.argtypes = [c_f64p, c_f64p]
        L.vo_extract_3d.argtypes = [c_f64p, c_f64p, c_f64p, c_f64p, c_f32p]
        L.vo_hm_pt_interp.argtypes = [c_f64p, C.c_int, C.c_double, C.c_double, C.c_double]
        L.vo_hm_pt_interp.restype = C.c_double
        L.vo_oef_create.argtypes = [C.c_double] * 4
        L.vo_oef_create.restype = C.c_void_p
        L.vo_oef_destroy.argtypes = [C.c_void_p]
        L.vo_oef_call.argtypes = [C.c_void_p, C.c_double, C.c_double, c_f64p]
        L.vo_est_create.argtypes = [C.c_void_p, c_f64p, C.c_int]
        L.vo_est_create.restype = C.c_void_p
        L.vo_est_destroy.argtypes = [C.c_void_p]
        L.vo_est_reset.argtypes = [C.c_void_p]
        L.vo_est_set_nep50.argtypes = [C.c_void_p, C.c_int]
        L.vo_est_postprocess.argtypes = [C.c_void_p, c_f32p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                         c_f64p, c_f32p]
        L.vo_est_infer.argtypes = [C.c_void_p, c_u8p, C.c_int, C.c_int, C.c_int64, C.c_double, C.c_double, c_f64p,
                                   c_f32p]
        # The blocked SGEMM stops scaling at ~16 threads (bench.py measured 8 / 16 / 32 / 64 / 128 -> 3.6 / 4.0 / 3.7 / 2.9 / 1.6
        # frames/s on the GPU box) and a 256-thread team on a box whose CPU share is 16 cores has taken 44 s per frame:
        # cap the team unless VNECT_ORACLE_THREADS says otherwise.  Results do not depend on the thread count.
        L.vo_set_threads.argtypes = [C.c_int]
        want = os.environ.get("VNECT_ORACLE_THREADS")
        L.vo_set_threads(int(want) if want else min(16, len(os.sched_getaffinity(0))))
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def cvround(v):
    return lib().vo_cvround(float(v))


def resize(img, f):
    """cv2.resize(img, (0,0), fx=f, fy=f, interpolation=INTER_LINEAR) restated (u8 / f32 / f64)."""
    img = np.ascontiguousarray(img)
    sh, sw = img.shape[:2]
    cn = 1 if img.ndim == 2 else img.shape[2]
    dh, dw = C.c_int(), C.c_int()
    lib().vo_resize_size(sh, sw, float(f), C.byref(dh), C.byref(dw))
    shape = (dh.value, dw.value) if img.ndim == 2 else (dh.value, dw.value, cn)
    out = np.empty(shape, img.dtype)
    if img.dtype == np.uint8:
        lib().vo_resize_u8(_p(img, c_u8p), sh, sw, cn, float(f), _p(out, c_u8p))
    elif img.dtype == np.float32:
        lib().vo_resize_f32(_p(img, c_f32p), sh, sw, cn, float(f), _p(out, c_f32p))
    elif img.dtype == np.float64:
        lib().vo_resize_f64(_p(img, c_f64p), sh, sw, cn, float(f), _p(out, c_f64p))
    else:
        raise TypeError(img.dtype)
    return out


def gen_input_batch(img, scales):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape[:2]
    sc = np.asarray(scales, dtype=np.float64)
    batch = np.empty((len(sc), 368, 368, 3), np.float32)
    scaler, ox, oy = C.c_double(), C.c_int(), C.c_int()
    rc = lib().vo_gen_input_batch(_p(img, c_u8p), H, W, W * 3, _p(sc, c_f64p), len(sc), _p(batch, c_f32p),
                                  C.byref(scaler), C.byref(ox), C.byref(oy))
    if rc:
        raise ValueError("gen_input_batch failed")
    return batch, scaler.value, [ox.value, oy.value]


def merge_scales(maps, scales):
    maps = np.ascontiguousarray(maps, dtype=np.float32)
    sc = np.asarray(scales, dtype=np.float64)
    avg = np.empty((4, 46, 46, 21), np.float64)
    lib().vo_merge_scales(_p(maps, c_f32p), _p(sc, c_f64p), len(sc), _p(avg, c_f64p))
    return avg


def extract_2d(hm_avg):
    hm_avg = np.ascontiguousarray(hm_avg, dtype=np.float64)
    out = np.empty((21, 2), np.float64)
    lib().vo_extract_2d(_p(hm_avg, c_f64p), _p(out, c_f64p))
    return out


def extract_3d(j2d, xm, ym, zm):
    j2d = np.ascontiguousarray(j2d, dtype=np.float64)
    xm, ym, zm = (np.ascontiguousarray(m, dtype=np.float64) for m in (xm, ym, zm))
    out = np.empty((21, 3), np.float32)
    lib().vo_extract_3d(_p(j2d, c_f64p), _p(xm, c_f64p), _p(ym, c_f64p), _p(zm, c_f64p), _p(out, c_f32p))
    return out


def hm_pt_interp(m, scale, point):
    m = np.ascontiguousarray(m, dtype=np.float64)
    return lib().vo_hm_pt_interp(_p(m, c_f64p), 1, float(scale), float(point[0]), float(point[1]))


class OneEuro:
    def __init__(self, freq, mincutoff=1.0, beta=0.0, dcutoff=1.0):
        self._h = lib().vo_oef_create(freq, mincutoff, beta, dcutoff)

    def __call__(self, x, timestamp):
        y = C.c_double()
        if lib().vo_oef_call(self._h, float(x), float(timestamp), C.byref(y)):
            raise ZeroDivisionError("float division by zero")
        return y.value

    def __del__(self):
        if getattr(self, "_h", None):
            lib().vo_oef_destroy(self._h)
            self._h = None


class Oracle:
    """The network half: weights in the reference schema -> (S,46,46,84) maps."""

    def __init__(self, weights, keep=False, paper_res2c=False):
        self._h = lib().vo_net_create()
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shp = (C.c_int64 * a.ndim)(*a.shape)
            if lib().vo_net_set_weight(self._h, name.encode(), _p(a, c_f32p), shp, a.ndim):
                raise ValueError(lib().vo_net_error(self._h).decode())
        lib().vo_net_options(self._h, int(keep), int(paper_res2c))

    def forward(self, batch):
        batch = np.ascontiguousarray(batch, dtype=np.float32)
        S = batch.shape[0]
        assert batch.shape[1:] == (368, 368, 3)
        out = np.empty((S, 46, 46, 84), np.float32)
        if lib().vo_net_forward(self._h, _p(batch, c_f32p), S, _p(out, c_f32p)):
            raise RuntimeError(lib().vo_net_error(self._h).decode())
        return out

    def activation(self, name):
        shp = (C.c_int * 4)()
        p = lib().vo_net_activation(self._h, name.encode(), shp)
        if not p:
            raise KeyError(name)
        n = shp[0] * shp[1] * shp[2] * shp[3]
        return np.ctypeslib.as_array(p, shape=(n,)).reshape(tuple(shp)).copy()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().vo_net_destroy(self._h)
            self._h = None


class OracleEstimator:
    """VNectEstimator.__call__ restated end to end (estimator.py:97-142) with injected timestamps."""

    def __init__(self, weights=None, scales=(1, 0.85, 0.7), nep50=False, net=None):
        self.net = net if net is not None else (Oracle(weights) if weights is not None else None)
        self.scales = [float(s) for s in scales]
        sc = np.asarray(self.scales, np.float64)
        self._h = lib().vo_est_create(self.net._h if self.net else None, _p(sc, c_f64p), len(sc))
        lib().vo_est_set_nep50(self._h, int(nep50))

    def reset(self):
        lib().vo_est_reset(self._h)

    def postprocess(self, maps, t2d, t3d, scaler=1.0, offset_x=0, offset_y=0):
        maps = np.ascontiguousarray(maps, dtype=np.float32)
        assert maps.shape == (len(self.scales), 46, 46, 84)
        j2, j3 = np.empty((21, 2), np.float64), np.empty((21, 3), np.float32)
        if lib().vo_est_postprocess(self._h, _p(maps, c_f32p), t2d, t3d, scaler, offset_x, offset_y, _p(j2, c_f64p),
                                    _p(j3, c_f32p)):
            raise ZeroDivisionError("float division by zero")
        return j2, j3

    def __call__(self, img, t2d, t3d):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        H, W = img.shape[:2]
        j2, j3 = np.empty((21, 2), np.float64), np.empty((21, 3), np.float32)
        if lib().vo_est_infer(self._h, _p(img, c_u8p), H, W, W * 3, t2d, t3d, _p(j2, c_f64p), _p(j3, c_f32p)):
            raise RuntimeError("oracle inference failed")
        return j2, j3

    def __del__(self):
        if getattr(self, "_h", None):
            lib().vo_est_destroy(self._h)
            self._h = None
