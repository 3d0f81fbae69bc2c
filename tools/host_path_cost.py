"""What does the PYTHON call path cost a synchronous frame?  bench.py times `Handle.infer_resident` (one call into the library + two fresh result arrays, as
the reference's __call__ returns fresh arrays); this runs the same loop (a) through that method, (b) through the bare ctypes function with the
result buffers reused, (c) from C (tools/c_loop_rate.c), in one process / one call, 2 000 frames each.  GPU box."""
import os, struct, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import helpers
from vnect_amd import _native
from vnect_amd.weights import synthetic_weights
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
w = synthetic_weights()
frames = [helpers.synth_frame(1234 + k) for k in range(8)]
blob = "/tmp/host_path_cost.bin"
with open(blob, "wb") as f:
    f.write(struct.pack("<i", len(w)))
    for name, arr in w.items():
        a = np.ascontiguousarray(arr, np.float32)
        f.write(struct.pack("<i", len(name)) + name.encode() + struct.pack("<i", a.ndim) + struct.pack("<%dq" % a.ndim, *a.shape) + a.tobytes())
    f.write(struct.pack("<iii", 8, 368, 368))
    for fr in frames:
        f.write(np.ascontiguousarray(fr).tobytes())
exe = os.path.join(ROOT, "tools", "c_loop_rate")
libdir = os.path.dirname(_native.LIB_PATH)
subprocess.check_call(["gcc", "-O2", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "c_loop_rate.c"), "-o", exe, "-L", libdir,
                       "-lvnect_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"])
for prec_name, prec in (("fp32", _native.FP32), ("bf16", _native.BF16)):
    h = _native.Handle([1.0, 0.8, 0.6], precision=prec, num_frame_slots=8, lanes=3)
    h.set_weights(w); h.finalize()
    for k in range(8):
        h.upload_frame(k, frames[k])
    N, t = 2000, 1.7e9
    res = {}
    for rep in range(2):
        for name in ("method", "bare"):
            j2, j3, p2, p3, fn, _ = h._results()
            lat = np.empty(N)
            for i in range(100 + N):
                t += 1 / 30
                a = time.perf_counter()
                if name != "bare":
                    h.infer_resident(i % 8, t, t + 1e-3)
                else:
                    fn(h._h, i % 8, t, t + 1e-3, p2, p3)
                if i >= 100:
                    lat[i - 100] = time.perf_counter() - a
            res[name] = float(np.median(lat)) * 1e3
    print("%s python: Handle.infer_resident median %.4f ms (%.1f frames/s); bare ctypes call, buffers reused %.4f ms (%.1f)" % (
        prec_name, res["method"], 1e3 / res["method"], res["bare"], 1e3 / res["bare"]), flush=True)
    h.close()
    print(subprocess.run([exe, blob] + (["bf16"] if prec_name == "bf16" else []), capture_output=True, text=True).stdout.strip(), flush=True)
