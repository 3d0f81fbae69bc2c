"""ctypes binding of libvnect_hip.so (include/vnect_abi.h).  No torch, no TensorFlow.

The library is the product path; there is no CPU fallback: if it is missing or no MI355X is
visible, loading / creating a handle raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VNECT_LIB") or os.path.join(_HERE, "lib", "libvnect_hip.so")  # VNECT_LIB: A/B tuning builds
_lib = None

MAX_SCALES = 8
OK, E_ARG, E_STATE, E_HIP, E_NODEVICE, E_TIMESTAMP, E_COMM, E_TIMEORDER, E_INTERNAL = 0, -1, -2, -3, -4, -5, -6, -7, -8
ABI_VERSION = 6
MAX_STREAMS = 4
XCHG_RCCL, XCHG_P2P = 0, 1
FP32, BF16, FP32_SPLIT = 0, 1, 2


class VnectError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libvnect_hip: %s (code %d)" % (msg, code))
        self.code = code


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("num_scales", C.c_int32),
                ("scales", C.c_double * MAX_SCALES), ("precision", C.c_int32), ("paper_res2c", C.c_int32),
                ("use_graph", C.c_int32), ("numpy_promotion", C.c_int32), ("max_frame_bytes", C.c_int32),
                ("num_frame_slots", C.c_int32), ("pyramid_nranks", C.c_int32), ("pyramid_rank", C.c_int32),
                ("keep_activations", C.c_int32), ("lanes", C.c_int32), ("preprocess_only", C.c_int32),
                ("exchange", C.c_int32)]


class Timings(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("frames", C.c_int32), ("total_ms", C.c_double), ("net_ms", C.c_double),
                ("conv_ms", C.c_double), ("conv_launches", C.c_int32), ("conv_flops", C.c_double), ("conv_slot_ms", C.c_double),
                ("shader_cycles", C.c_double), ("shader_ticks", C.c_double)]   # ABI v6: clock [MHz] = 100 * cycles / ticks


class LayerInfo(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("tile_m", C.c_int32), ("tile_n", C.c_int32), ("split_k", C.c_int32), ("workgroups", C.c_int32),
                ("flops", C.c_double), ("last_ms", C.c_double)]


# every symbol include/vnect_abi.h declares: name -> (restype, argtypes)
_f32p, _f64p, _u8p, _i32p = (C.POINTER(t) for t in (C.c_float, C.c_double, C.c_uint8, C.c_int32))
_H = C.c_void_p
SYMBOLS = {
    "vnect_abi_version": (C.c_int, []),
    "vnect_build_info": (C.c_char_p, []),
    "vnect_create": (C.c_int, [C.POINTER(Config), C.POINTER(_H)]),
    "vnect_destroy": (None, [_H]),
    "vnect_last_error": (C.c_char_p, [_H]),
    "vnect_set_weight": (C.c_int, [_H, C.c_char_p, _f32p, C.POINTER(C.c_int64), C.c_int]),
    "vnect_finalize": (C.c_int, [_H]),
    "vnect_set_scales": (C.c_int, [_H, _f64p, C.c_int]),
    "vnect_forward": (C.c_int, [_H, _f32p, C.c_int, _f32p]),
    "vnect_preprocess": (C.c_int, [_H, _u8p, C.c_int, C.c_int, C.c_int64, _f32p, _f64p, _i32p, _i32p]),
    "vnect_postprocess": (C.c_int, [_H, _f32p, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32, _f64p, _f32p]),
    "vnect_infer": (C.c_int, [_H, _u8p, C.c_int, C.c_int, C.c_int64, C.c_double, C.c_double, _f64p, _f32p]),
    "vnect_frame_buffer": (C.c_int, [_H, C.c_int, C.c_int64, C.POINTER(_u8p)]),
    "vnect_upload_frame": (C.c_int, [_H, C.c_int, _u8p, C.c_int, C.c_int, C.c_int64]),
    "vnect_infer_resident": (C.c_int, [_H, C.c_int, C.c_double, C.c_double, _f64p, _f32p]),
    "vnect_submit_resident": (C.c_int, [_H, C.c_int, C.c_double, C.c_double]),
    "vnect_collect": (C.c_int, [_H, _f64p, _f32p]),
    "vnect_submit_stream": (C.c_int, [_H, C.c_int, C.c_int, C.c_double, C.c_double]),
    "vnect_collect_stream": (C.c_int, [_H, _i32p, _f64p, _f32p]),
    "vnect_reset_filters_stream": (C.c_int, [_H, C.c_int]),
    "vnect_joint_filter": (C.c_int, [_H, C.c_int, _f64p, C.c_int, C.c_double, _f64p]),
    "vnect_reset_filters": (C.c_int, [_H]),
    "vnect_read_activation": (C.c_int, [_H, C.c_char_p, _f32p, C.c_int64, _i32p]),
    "vnect_set_profiling": (C.c_int, [_H, C.c_int]),
    "vnect_get_timings": (C.c_int, [_H, C.POINTER(Timings)]),
    "vnect_reset_timings": (C.c_int, [_H]),
    "vnect_get_layer_info": (C.c_int, [_H, C.c_int, C.POINTER(LayerInfo)]),
    "vnect_get_layer_stamps": (C.c_int, [_H, C.c_int, C.POINTER(C.c_uint64)]),
    "vnect_comm_unique_id": (C.c_int, [C.c_void_p]),
    "vnect_comm_init": (C.c_int, [_H, C.c_int, C.c_int, C.c_void_p]),
    "vnect_comm_library": (C.c_int, [C.c_char_p, C.c_int, _i32p]),
    "vnect_comm_p2p_export": (C.c_int, [_H, C.c_void_p]),
    "vnect_comm_p2p_init": (C.c_int, [_H, C.c_int, C.c_int, C.c_void_p]),
}


TESTHOOKS_LIB = os.path.join(_HERE, "lib", "libvnect_hip_testhooks.so")   # the host runtime with -DVNECT_TEST_HOOKS=1 (tests only)


def build(force=False):
    """hipcc --offload-arch=gfx950 build of the library (works without a GPU), and of its test twin: the same kernel objects under a
    host runtime compiled with the warm start's failure injection (`make testhooks`; only tests/test_gpu_surface.py loads it)."""
    src = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", src] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", src, "testhooks"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def build_info():
    """vnect_build_info() as a dict: {"abi": "6", "compiler": ..., "flags": ..., "variant": "", "test_hooks": "0", "probes_off": "1",
    "conv": "X3_DBG=0 ...", "post": "..."}."""
    text = lib().vnect_build_info().decode()
    out = {"text": text}
    for part in text.split("; "):
        if ": " in part and "=" not in part.split(": ", 1)[0]:
            k, v = part.split(": ", 1)
        else:
            k, v = part.split("=", 1)
        out[k.strip()] = v.strip()
    return out


def lib():
    """Load the library; raises if it has not been built (there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libvnect_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(needs hipcc); the VNect path has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the ABI and the header drift apart
            fn.restype, fn.argtypes = res, args
        if L.vnect_abi_version() != ABI_VERSION:
            raise ImportError("libvnect_hip.so ABI version mismatch")
        _lib = L
    return _lib


def _ptr(a, t):
    return a.ctypes.data_as(t)


class Handle:
    """Thin RAII wrapper over vnect_handle; every method maps 1:1 to a C entry point."""

    def __init__(self, scales, device=0, precision=FP32, paper_res2c=False, use_graph="auto", numpy_promotion=0,
                 max_frame_bytes=0, num_frame_slots=0, pyramid=None, keep_activations=False, lanes=1,
                 preprocess_only=False, exchange=XCHG_RCCL):
        L = lib()
        cfg = Config()
        cfg.struct_size = C.sizeof(Config)
        cfg.device = device
        cfg.num_scales = len(scales)
        for i, s in enumerate(scales):
            cfg.scales[i] = float(s)
        # use_graph: False = eager launches, True = always replay the frame graph, "auto" = eager for a synchronous frame,
        # graph replay when frames are in flight (the default)
        cfg.precision, cfg.paper_res2c, cfg.use_graph = precision, int(paper_res2c), (2 if use_graph == "auto" else int(bool(use_graph)))
        cfg.numpy_promotion, cfg.max_frame_bytes, cfg.num_frame_slots = numpy_promotion, max_frame_bytes, num_frame_slots
        cfg.lanes = int(lanes)  # 2: submit_resident/collect overlap two frames on two lanes
        cfg.keep_activations = int(keep_activations)  # True: activation(name) can return inner layers (tests)
        cfg.preprocess_only = int(preprocess_only)    # gen_input_batch only: no weights, no launch plan
        cfg.exchange = int(exchange)                  # pyramid sharding: XCHG_RCCL (ncclAllGather) or XCHG_P2P (peer writes)
        if pyramid is not None:  # (rank, nranks): this handle runs one scale of the pyramid (vnect_comm_init)
            cfg.pyramid_rank, cfg.pyramid_nranks = int(pyramid[0]), int(pyramid[1])
        h = _H()
        rc = L.vnect_create(C.byref(cfg), C.byref(h))
        self._h = h if h.value else None
        self.num_scales = len(scales)
        self.net_images = 1 if pyramid is not None else len(scales)
        if rc:
            msg = L.vnect_last_error(self._h).decode()
            self.close()
            raise VnectError(rc, msg)

    def close(self):
        if getattr(self, "_h", None):
            lib().vnect_destroy(self._h)
            self._h = None

    __del__ = close

    def _ck(self, rc):
        if rc:
            raise VnectError(rc, lib().vnect_last_error(self._h).decode())

    def set_weights(self, weights):
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shp = (C.c_int64 * a.ndim)(*a.shape)
            self._ck(lib().vnect_set_weight(self._h, name.encode(), _ptr(a, _f32p), shp, a.ndim))

    def finalize(self):
        self._ck(lib().vnect_finalize(self._h))

    def set_scales(self, scales):
        sc = np.asarray(scales, np.float64)
        self._ck(lib().vnect_set_scales(self._h, _ptr(sc, _f64p), len(sc)))

    def forward(self, batch):
        batch = np.ascontiguousarray(batch, dtype=np.float32)
        if batch.shape != (self.net_images, 368, 368, 3):
            raise ValueError("batch must be (%d,368,368,3)" % self.net_images)
        out = np.empty((self.net_images, 46, 46, 84), np.float32)
        self._ck(lib().vnect_forward(self._h, _ptr(batch, _f32p), self.net_images, _ptr(out, _f32p)))
        return out

    def preprocess(self, img, want_batch=True):
        img = _as_frame(img)
        H, W = img.shape[:2]
        batch = np.empty((self.net_images, 368, 368, 3), np.float32) if want_batch else None
        scaler, ox, oy = C.c_double(), C.c_int32(), C.c_int32()
        self._ck(lib().vnect_preprocess(self._h, _ptr(img, _u8p), H, W, img.strides[0],
                                        _ptr(batch, _f32p) if want_batch else None, C.byref(scaler), C.byref(ox),
                                        C.byref(oy)))
        return batch, scaler.value, [ox.value, oy.value]

    def postprocess(self, maps, t2d, t3d, scaler=1.0, offset_x=0, offset_y=0):
        maps = np.ascontiguousarray(maps, dtype=np.float32)
        if maps.shape != (self.num_scales, 46, 46, 84):
            raise ValueError("maps must be (%d,46,46,84)" % self.num_scales)
        j2, j3 = np.empty((21, 2), np.float64), np.empty((21, 3), np.float32)
        self._ck(lib().vnect_postprocess(self._h, _ptr(maps, _f32p), t2d, t3d, scaler, offset_x, offset_y,
                                         _ptr(j2, _f64p), _ptr(j3, _f32p)))
        return j2, j3

    def infer(self, img, t2d, t3d):
        img = _as_frame(img)
        H, W = img.shape[:2]
        j2, j3 = np.empty((21, 2), np.float64), np.empty((21, 3), np.float32)
        self._ck(lib().vnect_infer(self._h, _ptr(img, _u8p), H, W, img.strides[0], t2d, t3d, _ptr(j2, _f64p),
                                   _ptr(j3, _f32p)))
        return j2, j3

    def frame_buffer(self, index, H, W):
        """(H, W, 3) uint8 array over the handle's pinned staging buffer `index` (0 / 1): frames captured into it -- or crops of them --
        go to the device without a CPU copy when passed to infer().  Valid until a larger request for the same index or close()."""
        p = _u8p()
        self._ck(lib().vnect_frame_buffer(self._h, index, H * W * 3, C.byref(p)))
        return np.ctypeslib.as_array(p, shape=(H, W, 3))

    def upload_frame(self, slot, img):
        img = _as_frame(img)
        H, W = img.shape[:2]
        self._ck(lib().vnect_upload_frame(self._h, slot, _ptr(img, _u8p), H, W, img.strides[0]))

    # The per-frame calls sit between two frames of a synchronous stream (the GPU idles meanwhile): they hand the library
    # one pair of result buffers whose ctypes pointers are made once, and return fresh copies as the reference does.
    def _results(self):
        if getattr(self, "_res", None) is None:
            j2, j3 = np.empty((21, 2), np.float64), np.empty((21, 3), np.float32)
            self._res = (j2, j3, _ptr(j2, _f64p), _ptr(j3, _f32p), lib().vnect_infer_resident, lib().vnect_collect)
        return self._res

    def infer_resident(self, slot, t2d, t3d):
        j2, j3, p2, p3, fn, _ = self._results()
        rc = fn(self._h, slot, t2d, t3d, p2, p3)
        if rc:
            self._ck(rc)
        return j2.copy(), j3.copy()

    def submit_resident(self, slot, t2d, t3d):
        self._ck(lib().vnect_submit_resident(self._h, slot, t2d, t3d))

    def collect(self):
        j2, j3, p2, p3, _, fn = self._results()
        rc = fn(self._h, p2, p3)
        if rc:
            self._ck(rc)
        return j2.copy(), j3.copy()

    def submit_stream(self, stream, slot, t2d, t3d):
        """One frame of video stream `stream` (its own filter bank and timestamps on this handle): frames of different streams
        overlap on the handle's lanes with no dependency between them."""
        self._ck(lib().vnect_submit_stream(self._h, stream, slot, t2d, t3d))

    def collect_stream(self):
        """(stream, joints_2d, joints_3d) of the oldest frame in flight."""
        j2, j3, p2, p3, _, _ = self._results()
        s = C.c_int32(-1)
        rc = lib().vnect_collect_stream(self._h, C.byref(s), p2, p3)
        if rc:
            self._ck(rc)
        return s.value, j2.copy(), j3.copy()

    @staticmethod
    def comm_unique_id():
        buf = (C.c_char * 128)()
        rc = lib().vnect_comm_unique_id(buf)
        if rc:
            raise VnectError(rc, lib().vnect_last_error(None).decode())
        return bytes(buf)

    @staticmethod
    def comm_library():
        """(path of the RCCL the library resolved, True if it reused a copy the process had mapped already -- torch's)."""
        buf, reused = C.create_string_buffer(1024), C.c_int32(-1)
        rc = lib().vnect_comm_library(buf, len(buf), C.byref(reused))
        if rc:
            raise VnectError(rc, lib().vnect_last_error(None).decode())
        return buf.value.decode(), bool(reused.value)

    def comm_init(self, rank, nranks, unique_id):
        buf = (C.c_char * 128).from_buffer_copy(unique_id)
        self._ck(lib().vnect_comm_init(self._h, rank, nranks, buf))

    def joint_filter(self, dim, values, values_are_f32, t):
        """The handle's 2-D / 3-D OneEuro bank over (21, dim) values at timestamp t -> filtered float64 array."""
        vin = np.ascontiguousarray(values, dtype=np.float64)
        if vin.shape != (21, dim):
            raise ValueError("joints must be (21, %d)" % dim)
        out = np.empty((21, dim), np.float64)
        self._ck(lib().vnect_joint_filter(self._h, dim, _ptr(vin, _f64p), int(values_are_f32), float(t), _ptr(out, _f64p)))
        return out

    def p2p_export(self):
        """128-byte description of this rank's exchange block (exchange=XCHG_P2P); give every rank everybody's."""
        buf = (C.c_char * 128)()
        self._ck(lib().vnect_comm_p2p_export(self._h, buf))
        return bytes(buf)

    def p2p_init(self, rank, nranks, blobs):
        raw = b"".join(blobs)
        if len(raw) != 128 * nranks:
            raise ValueError("p2p_init needs one 128-byte blob per rank")
        buf = (C.c_char * len(raw)).from_buffer_copy(raw)
        self._ck(lib().vnect_comm_p2p_init(self._h, rank, nranks, buf))

    def reset_filters_stream(self, stream):
        self._ck(lib().vnect_reset_filters_stream(self._h, stream))

    def reset_filters(self):
        self._ck(lib().vnect_reset_filters(self._h))

    def activation(self, name):
        shp = (C.c_int32 * 4)()
        self._ck(lib().vnect_read_activation(self._h, name.encode(), None, 0, shp))
        out = np.empty(tuple(shp), np.float32)
        self._ck(lib().vnect_read_activation(self._h, name.encode(), _ptr(out, _f32p), out.size, shp))
        return out

    def set_profiling(self, on):
        self._ck(lib().vnect_set_profiling(self._h, int(on)))

    def timings(self):
        t = Timings()
        t.struct_size = C.sizeof(Timings)
        self._ck(lib().vnect_get_timings(self._h, C.byref(t)))
        d = {k: getattr(t, k) for k, _ in Timings._fields_ if k != "struct_size"}
        # the shader clock the chip held while the conv launches of the profiled frames ran (s_memtime / s_memrealtime of workgroup 0)
        d["shader_clock_mhz"] = 100.0 * d["shader_cycles"] / d["shader_ticks"] if d["shader_ticks"] > 0 else None
        return d

    def reset_timings(self):
        self._ck(lib().vnect_reset_timings(self._h))

    def layer_stamps(self, idx):
        """Raw device-clock stamps (100 MHz) of layer idx in the last profiled frame; see vnect_get_layer_stamps."""
        buf = (C.c_uint64 * 24)()
        self._ck(lib().vnect_get_layer_stamps(self._h, idx, buf))
        return list(buf)

    def layers(self):
        out, i = [], 0
        while True:
            li = LayerInfo()
            if lib().vnect_get_layer_info(self._h, i, C.byref(li)):
                return out
            out.append({k: (getattr(li, k).decode() if k == "name" else getattr(li, k)) for k, _ in LayerInfo._fields_})
            i += 1


def _as_frame(img):
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
        raise ValueError("frame must be a uint8 (H, W, 3) BGR array")
    # A positive row stride may stay arbitrary (crops of a larger frame are passed by stride, no copy); anything else the
    # reference accepts -- flipped views (negative strides), broadcast rows (stride 0), channel-swapped views -- is copied
    if img.strides[2] != 1 or img.strides[1] != 3 or img.strides[0] < img.shape[1] * 3:
        img = np.ascontiguousarray(img)
    return img
