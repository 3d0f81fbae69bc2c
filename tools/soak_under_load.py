"""The soak again under UNEVEN load (the guide's rule for every hand-off: test it with the chip busy and the consumer warm, checking every word).
post_kernel's relaxed `sc1` hand-off -- 168 arg-max workgroups publish partials, the last arriver runs the joints stage -- and the streaming conv
kernel's in-workgroup flags are exercised while ANOTHER handle (other frames, the other precision, its own host thread) hammers the same GPU with
frames three deep, and a third thread streams 64 MB device-to-device copies.  The handle under test runs 10 000 synchronous frames; every frame's joints
must equal, bit for bit, the same sequence run alone.  fp32 and bf16.  GPU box, ~2 minutes."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests import helpers
from vnect_amd import _native
from vnect_amd.weights import synthetic_weights
w = synthetic_weights()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
for prec, other, name in ((_native.FP32, _native.BF16, "fp32"), (_native.BF16, _native.FP32, "bf16")):
    h = _native.Handle([1.0, 0.8, 0.6], precision=prec, num_frame_slots=4)
    h.set_weights(w); h.finalize()
    for k in range(4):
        h.upload_frame(k, helpers.synth_frame(100 + k, smooth=True))

    def sequence():
        h.reset_filters()
        out2, out3 = np.empty((N, 21, 2)), np.empty((N, 21, 3), np.float32)
        for i in range(N):
            out2[i], out3[i] = h.infer_resident(i % 4, 10.0 + i / 30, 10.0 + i / 30 + 1e-3)
        return out2, out3

    t0 = time.time()
    alone = sequence()
    t_alone = time.time() - t0
    # the load: another handle three frames deep + a copy stream
    hb = _native.Handle([1.0, 0.8, 0.6], precision=other, lanes=3, num_frame_slots=4)
    hb.set_weights(w); hb.finalize()
    for k in range(4):
        hb.upload_frame(k, helpers.synth_frame(500 + k))
    stop = threading.Event()
    load_frames = [0]

    def hammer():
        i = 0
        while not stop.is_set():
            if i >= 3:
                hb.collect()
            hb.submit_resident(i % 4, 20.0 + i / 30, 20.0 + i / 30 + 1e-3)
            i += 1
        for _ in range(min(3, i)):
            hb.collect()
        load_frames[0] = i

    def copier():
        a = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
        b = torch.empty_like(a)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            while not stop.is_set():
                for _ in range(8):
                    b.copy_(a, non_blocking=True)
                s.synchronize()

    ths = [threading.Thread(target=hammer), threading.Thread(target=copier)]
    for th in ths:
        th.start()
    t0 = time.time()
    loaded = sequence()
    t_loaded = time.time() - t0
    stop.set()
    for th in ths:
        th.join()
    bad2 = int(np.any(alone[0] != loaded[0], axis=(1, 2)).sum())
    bad3 = int(np.any(alone[1] != loaded[1], axis=(1, 2)).sum())
    print("%s under load: %d frames, %d differ in joints_2d, %d in joints_3d; alone %.1f s, beside %d frames of the other handle + the copy stream %.1f s"
          % (name, N, bad2, bad3, t_alone, load_frames[0], t_loaded), flush=True)
    assert bad2 == 0 and bad3 == 0
    hb.close()
    h.close()
print("soak under load ok")
