"""Frame time of 1- and 2-scale handles with and without the wide tail / chain forms (are they right for FEWER workgroups than CUs?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers
from vnect_amd import _native
from vnect_amd.weights import synthetic_weights
W = synthetic_weights()
for prec, name in ((_native.FP32, "fp32"), (_native.BF16, "bf16")):
    for scales in ([1.0], [1.0, 0.7]):
        for tag, env in (("default", {}), ("no wide tail", {"VNECT_NO_WIDE_TAIL": "1"}), ("force wide tail", {"VNECT_FORCE_WIDE_TAIL": "1"}), ("no tails", {"VNECT_NO_TAIL": "1"})):
            os.environ.update(env)
            h = _native.Handle(scales, precision=prec)
            h.set_weights(W); h.finalize()
            for k in env: del os.environ[k]
            h.upload_frame(0, helpers.synth_frame(1234))
            for i in range(30):
                h.infer_resident(0, 1.0 + i, 1.0 + i)
            n = 300
            t0 = time.perf_counter()
            for i in range(n):
                h.infer_resident(0, 100.0 + i, 100.0 + i)
            dt = time.perf_counter() - t0
            print("%s scales %-10s %-13s %.3f ms per frame (%.0f frames/s)" % (name, scales, tag, dt / n * 1e3, n / dt))
            h.close()
