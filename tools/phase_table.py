"""Per-layer timeline of the conv kernels from the device-clock phase stamps (tuning aid; runs on the GPU box).

Columns (us, medians over frames): kernel = earliest workgroup start .. latest workgroup end; gap = previous conv kernel's
end .. this kernel's start (includes pool / reduce kernels where present); for workgroup 0: setup (start .. operands
requested), land (.. first chunk in LDS), loop (.. K loop done), epi (.. stores done); lastwg = start of the
last-dispatched workgroups relative to the kernel start; p.wait / p.bar / p.iss = time producer wave 0 spent waiting for
landings / at the barrier / issuing LDS-DMA, c.bar = time consumer wave 0 waited at the per-chunk barrier (at ~2.3 GHz).
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import helpers
from vnect_amd import _native
from vnect_amd.weights import synthetic_weights

h = _native.Handle([1.0, 0.8, 0.6], precision=_native.BF16 if os.environ.get("LT_BF16") == "1" else _native.FP32)
h.set_weights(synthetic_weights()); h.finalize()
h.upload_frame(0, helpers.synth_frame(1234))
for i in range(5):
    h.infer_resident(0, 1.0 + i, 1.0 + i)
h.set_profiling(True)
N = 15
rows = {}
names = [l for l in h.layers()]
for f in range(N):
    h.infer_resident(0, 10.0 + f, 10.0 + f)
    prev_end = None
    for i, l in enumerate(names):
        st = h.layer_stamps(i)
        if st[0] == 2**64 - 1 or max(st[1:9]) == 0:
            continue
        t0, t1 = st[0], max(st[1:9])
        r = [(t1 - t0), (t0 - prev_end) if prev_end else 0, st[10] - st[9], st[11] - st[10], st[12] - st[11], st[13] - st[12],
             st[15] - t0, st[9] - t0, st[16] / 23.0, st[17] / 23.0, st[18] / 23.0, st[19] / 23.0]
        if f == N - 1 and os.environ.get('PT_PROD'):
            print(l['name'][:24], 'producer wave 0 from workgroup start: args pinned %.2f, first item decoded %.2f, ring requested %.2f, chunk 0 landed %.2f us' % tuple((st[k] - st[9]) / 100.0 for k in (22, 23, 14, 11)))
        if f == N - 1 and os.environ.get('PT_SETUP'):
            print(l['name'][:24], 'from first chunk landed: K loop of item 0 %.2f, +epilogue issued %.2f, all items %.2f, stores drained %.2f' % tuple((st[k] - st[11]) / 100.0 for k in (20, 21, 12, 13)))
        rows.setdefault(i, []).append(r)
        prev_end = t1
print("%-30s %5s %3s %5s | %7s %6s | %6s %6s %6s %6s | %6s %6s | %6s %6s %6s %6s" % ("layer", "WGs", "ks", "chunk", "kernel", "gap", "setup", "land", "loop", "epi", "lastwg", "wg0", "p.wait", "p.bar", "p.iss", "c.bar"))
tot = np.zeros(12)
for i, l in enumerate(names):
    if i not in rows:
        continue
    m = np.median(np.array(rows[i], dtype=np.float64), axis=0) / 100.0
    tot += m
    ch = (l["K"] + 31) // 32 // max(1, l["split_k"])
    print("%-30s %5d %3d %5d | %7.2f %6.2f | %6.2f %6.2f %6.2f %6.2f | %6.2f %6.2f | %6.2f %6.2f %6.2f %6.2f" % (l["name"][:30], l["workgroups"], l["split_k"], ch, *m))
print("%-30s %5s %3s %5s | %7.2f %6.2f | %6.2f %6.2f %6.2f %6.2f | %6.2f %6.2f | %6.2f %6.2f %6.2f %6.2f" % ("sum", "", "", "", *tot))
