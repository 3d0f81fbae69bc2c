"""How often does a heat-map of the synthetic-weight net have a REAL maximum -- one bf16 noise cannot move?  (tests/test_gpu_bf16.py::
test_bf16_margin_conditioned_joints needs such (frame, joint) pairs to exist.)  For a few weight variants and scale sets: the margin of every
fp32 maximum in units of eps = 3e-2 * max|fp32 maps| (the bf16 map gate), (a) in the x8-upsampled map against the best pixel outside the
maximum's 8x8 block, (b) in the merged 46x46 map against the best other cell.  Prints counts; writes nothing.

    python3 tools/bf16_margin_probe.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from tests import helpers
from vnect_amd import _native
from vnect_amd.weights import synthetic_weights


def variants():
    w = synthetic_weights()
    yield "default", w
    # heat-map columns of the last 1x1 layer scaled up: `top` is then a heat-map value, eps is relative to the heat-maps' own range
    for c in (4.0,):
        v = dict(w)
        k = v["res5c_branch2c/kernel"].copy()      # (1,1,128,84): columns [0,21) are the heat-maps
        k[..., :21] *= c
        v["res5c_branch2c/kernel"] = k
        yield "heat columns x%g" % c, v
    # heat-map j = ONE feature channel of res5c_branch2b (post-ReLU), scaled so that the heat-maps dominate
    v = dict(w)
    k = v["res5c_branch2c/kernel"].copy()
    k[..., :21] = 0
    for j in range(21):
        k[0, 0, 6 * j, j] = 2.0
    v["res5c_branch2c/kernel"] = k
    yield "heat-map j = feature 6j x2", v
    # the same with a negative bias on res5c_branch2b: sparse features, isolated peaks
    for q in (0.5, 1.0):
        v2 = dict(v)
        b = v2["res5c_branch2b/biases"].copy()
        v2["res5c_branch2b/biases"] = b - q
        yield "one-hot + res5c_branch2b bias - %g" % q, v2


def margins(avg, raw):
    up = oracle.resize(np.ascontiguousarray(avg), 8.0)
    r0, c0 = (int(raw[0]) // 8) * 8, (int(raw[1]) // 8) * 8
    rest = up.copy()
    rest[r0:r0 + 8, c0:c0 + 8] = -np.inf
    m_up = float(up[int(raw[0]), int(raw[1])] - rest.max())
    flat = np.sort(avg.ravel())
    return m_up, float(flat[-1] - flat[-2])


shapes = [(368, 368), (538, 368), (240, 320), (368, 300)]
for scales in ([1.0, 0.8, 0.6], [1.0]):
    for name, w in variants():
        hf = _native.Handle(scales); hf.set_weights(w); hf.finalize()
        hb = _native.Handle(scales, precision=_native.BF16); hb.set_weights(w); hb.finalize()
        n_up = n_cell = moved_up = moved_cell = 0
        errs, mu, mc = [], [], []
        for k in range(12):
            H, W = shapes[k % 4]
            frame = helpers.synth_frame(91000 + k, H, W, smooth=True)
            hf.reset_filters(), hb.reset_filters()
            hf.infer(frame, 5.0, 5.0), hb.infer(frame, 5.0, 5.0)
            mf, mb = hf.activation("res5c_branch2c"), hb.activation("res5c_branch2c")
            top = float(np.abs(mf).max())
            eps = 3e-2 * top
            errs.append(float(np.abs(mb - mf).max()) / top)
            af, ab = oracle.merge_scales(mf, scales)[0], oracle.merge_scales(mb, scales)[0]
            rf, rb = oracle.extract_2d(af), oracle.extract_2d(ab)
            for j in range(21):
                a, b = margins(af[:, :, j], rf[j])
                mu.append(a / eps), mc.append(b / eps)
                same = bool(np.all(rb[j] // 8 == rf[j] // 8))
                if a > 2 * eps:
                    n_up += 1
                    moved_up += not same
                if b > 2 * eps:
                    n_cell += 1
                    moved_cell += not same
        mu, mc = np.array(mu), np.array(mc)
        print("scales %-16s %-40s bf16 map err %.3g..%.3g of max | upsampled-block margin > 2 eps: %3d of %d (moved %d), p50 %.2f p90 %.2f max %.2f eps | "
              "cell margin > 2 eps: %3d (moved %d), p90 %.2f max %.2f eps"
              % (scales, name, min(errs), max(errs), n_up, len(mu), moved_up, np.median(mu), np.percentile(mu, 90), mu.max(), n_cell, moved_cell,
                 np.percentile(mc, 90), mc.max()), flush=True)
        hf.close(), hb.close()
