"""GPU (MI355X), through the C ABI: the split-product fp32 path (VNECT_FP32_SPLIT; frozen secondary leg), held to the fp32 gates."""
import json
import os

import numpy as np
import pytest

from tests.gpu_common import BASELINE_SCALES, G, OUT, T0, _EndToEnd, _handle, _log, _native, _round_bf16  # noqa: F401

pytestmark = pytest.mark.gpu


def test_split_product_path_meets_the_fp32_gates(weights, oracle_net):
    """precision = FP32_SPLIT: fp32 tensors and accumulators, but the 64x64-tile layers form their products on the bf16 matrix pipe from
    exact three-way splits of both operands (6 of the 9 piece products; conv.hip, X3).  It is an fp32-class path, so it is held to
    the FP32 gates, not the bf16 ones: every layer <= 1e-4 * max|oracle| (its error is printed beside the fp32 instruction's), arena
    plan == private-buffer plan bit for bit, and the whole __call__ through the every-frame, every-joint gate of _EndToEnd."""
    import oracle
    from tests import helpers
    n = _native()
    batch, _, _ = oracle.gen_input_batch(helpers.synth_frame(1234, smooth=True), BASELINE_SCALES)
    ref = oracle_net.forward(batch)
    hs = _handle(BASELINE_SCALES, weights, precision=n.FP32_SPLIT, keep_activations=True)
    hf = _handle(BASELINE_SCALES, weights, keep_activations=True)
    out_s, out_f = hs.forward(batch), hf.forward(batch)
    rows = []
    for name in ["pool1", "res2a_branch2a", "res2a", "res2c", "res3a", "res3d", "res4a_branch2b", "res4f", "res5a_branch2a_new", "res5a",
                 "res5b_branch2c_new", "res5c_branch2a_feat", "res5c_branch2b", "res5c_branch2c"]:
        r = oracle_net.activation(name)
        top = float(np.abs(r).max())
        es, ef = float(np.abs(hs.activation(name) - r).max()) / top, float(np.abs(hf.activation(name) - r).max()) / top
        rows.append((name, es, ef))
        print("%-24s split %.3g   fp32 instruction %.3g" % (name, es, ef))
    _log("split_layer_errors.json", rows)
    assert all(es <= 1e-4 for _, es, _ in rows), rows
    assert float(np.abs(out_s - ref).max()) <= 1e-4 * float(np.abs(ref).max())
    hs.close(), hf.close()
    arena = _handle(BASELINE_SCALES, weights, precision=n.FP32_SPLIT)
    assert any(">" in L["name"] for L in arena.layers())           # the tail-fused launches are in this plan
    # (the arena plan's tail GEMMs multiply with the fp32 instruction, the private-buffer plan's stand-alone 1x1 layers by splits: equal
    # to fp32 rounding, not bit for bit -- in this mode the last bits depend on the launch plan, like any change of summation order)
    assert float(np.abs(arena.forward(batch) - out_s).max()) <= 2e-5 * float(np.abs(ref).max())
    e2e = _EndToEnd(BASELINE_SCALES, oracle_net)
    for k, (H, W) in enumerate([(368, 368), (538, 368), (240, 320), (368, 368)]):
        frame = helpers.synth_frame(8800 + k, H, W, smooth=True)
        t = T0 + 300 + k / 30
        j2, j3 = arena.infer(frame, t, t + 0.001)
        e2e.check(frame, t, t + 0.001, j2, j3, arena.activation("res5c_branch2c"), (H, W))
    print("split-product path: legal arg-max ties %d, worst 3-D excess over tolerance %.3g" % (e2e.ties, e2e.worst3))
    arena.close()


def test_split_product_path_vs_float64(weights):
    """Is the split-product path "reduced precision"?  Both GPU paths against the TRUE result: the torch float64 restatement of the
    graph (tests/torch_net.py, independent of the C oracle) on one image.  The split-product maps must be as close to float64 as the
    fp32 instruction's are (within 1.5x of its max error and of its RMS error), and both far inside the fp32 gate."""
    import oracle
    import torch
    from tests import helpers, torch_net
    n = _native()
    torch.set_num_threads(16)
    batch, _, _ = oracle.gen_input_batch(helpers.synth_frame(4242, smooth=True), [1.0])
    with torch.inference_mode():
        ref = torch_net.forward(weights, batch, dtype=torch.float64).numpy()
    top = float(np.abs(ref).max())
    errs = {}
    for name, prec in (("fp32 instruction", n.FP32), ("split product", n.FP32_SPLIT)):
        h = _handle([1.0], weights, precision=prec)
        d = h.forward(batch).astype(np.float64) - ref
        h.close()
        errs[name] = (float(np.abs(d).max()) / top, float(np.sqrt((d * d).mean())) / top)
        print("%-18s vs float64: max %.3g  rms %.3g (of max|map|)" % (name, *errs[name]))
    f, s = errs["fp32 instruction"], errs["split product"]
    assert s[0] <= 1.5 * f[0] and s[1] <= 1.5 * f[1], errs
    assert s[0] <= 2e-5 and f[0] <= 2e-5
