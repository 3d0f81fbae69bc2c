"""What ONE pyramid rank has to do per frame (SURVEY 8e, configs[3]): the synchronous frame rate of a single-scale handle, per scale set and
precision, on one GPU.  A pyramid-sharded rank runs this plan (S = 1) plus the exchange; three ranks on three GPUs cannot be faster per frame
than this.  Not a measurement of configs[3].  `python3 tools/one_scale_rate.py [out.json]` also writes the numbers as JSON
(committed as profiles/rNN_one_scale_rate.json: bench.py's --pyramid line cites the newest one instead of carrying literals)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers
from vnect_amd import _native
from vnect_amd.weights import synthetic_weights
W = synthetic_weights()
res = {}
for prec, name in ((_native.FP32, "fp32"), (_native.BF16, "bf16")):
    for scales in ([1.0], [1.0, 0.8, 0.6]):
        h = _native.Handle(scales, precision=prec)
        h.set_weights(W); h.finalize()
        h.upload_frame(0, helpers.synth_frame(1234))
        for i in range(30):
            h.infer_resident(0, 1.0 + i, 1.0 + i)
        n = 300
        t0 = time.perf_counter()
        for i in range(n):
            h.infer_resident(0, 100.0 + i, 100.0 + i)
        dt = time.perf_counter() - t0
        print("%s scales %s: %.1f frames/s, %.3f ms per frame, %d conv launches" % (name, scales, n / dt, dt / n * 1e3, h.timings()["conv_launches"] if "conv_launches" in h.timings() else -1))
        res["%s_%dscale" % (name, len(scales))] = round(dt / n * 1e3, 4)
        h.close()
if len(sys.argv) > 1:
    json.dump({"ms_per_frame": res, "frames": 300, "warmup": 30, "what": "synchronous frames of a single-scale handle (what ONE pyramid rank "
               "computes per frame, without its exchange) against the three-scale handle on the same GPU, same call"}, open(sys.argv[1], "w"), indent=1)
