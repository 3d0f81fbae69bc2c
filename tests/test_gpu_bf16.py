"""GPU (MI355X), through the C ABI: the bf16 MFMA conv path (BASELINE.json configs[2]), tolerance-gated against fp32."""
import json
import os

import numpy as np
import pytest

from tests.gpu_common import BASELINE_SCALES, G, OUT, T0, _EndToEnd, _handle, _log, _native, _round_bf16  # noqa: F401

pytestmark = pytest.mark.gpu


def test_bf16_path_gated_against_fp32(weights, oracle_net, h3):
    """bf16 MFMA conv path: activations and weights are bf16 (8 significant bits), accumulation fp32, final maps
    and post-processing fp32/f64.  Gate (calibrated on MI355X, seeded synthetic weights, 54 layers deep):
    every layer <= 4e-2 * max|ref|, final maps <= 3e-2 * max|ref|; the fp32 path sits at 3e-6 on the same input."""
    import oracle
    from tests import helpers
    batch, _, _ = oracle.gen_input_batch(helpers.synth_frame(1234, smooth=True), BASELINE_SCALES)
    ref = oracle_net.forward(batch)
    hb = _handle(BASELINE_SCALES, weights, precision=_native().BF16, keep_activations=True)
    out = hb.forward(batch)
    f32 = h3.forward(batch)
    rows = []
    for n in ["conv1", "pool1", "res2a", "res2c", "res3d", "res4a_branch2b", "res4f", "res5a", "res5b_branch2c_new",
              "res5c_branch2a_feat", "res5c_branch2b", "res5c_branch2c"]:
        a, r = hb.activation(n), oracle_net.activation(n)
        assert a.shape == r.shape, n
        rows.append((n, float(np.abs(a - r).max() / np.abs(r).max())))
    _log("bf16_layer_errors.json", rows)
    for n, e in rows:
        print("%-24s rel err %.3g" % (n, e))
    e_out = float(np.abs(out - ref).max() / np.abs(ref).max())
    e_f32 = float(np.abs(f32 - ref).max() / np.abs(ref).max())
    print("final maps: bf16 %.3g, fp32 %.3g" % (e_out, e_f32))
    assert all(e <= 4e-2 for _, e in rows), rows
    assert e_out <= 3e-2 and e_f32 <= 1e-4
    # the preprocessing is the same integer arithmetic, rounded once to bf16 at the end
    frame = helpers.synth_frame(9, 300, 420, smooth=True)
    bb, s, off = hb.preprocess(frame)
    rb, rs, roff = oracle.gen_input_batch(frame, BASELINE_SCALES)
    assert s == rs and off == roff and np.abs(bb - rb).max() <= 2 ** -8
    # end to end: joints from planted-peak-free noise maps are tie-prone; require finite, well-formed output and
    # agreement of most joints within one heat-map cell (8 px) with the fp32 path
    j2b, j3b = hb.infer(frame, T0, T0 + 0.001)
    h3.reset_filters()
    j2f, j3f = h3.infer(frame, T0, T0 + 0.001)
    assert np.all(np.isfinite(j2b)) and np.all(np.isfinite(j3b))
    close = np.all(np.abs(j2b - j2f) <= 8.0 / min(s, 1.0) + 1e-9, axis=1)
    same = np.all(j2b == j2f, axis=1)
    mb, mf = hb.activation("res5c_branch2c"), h3.activation("res5c_branch2c")
    # 3-D read-off where both paths sit on the same pixel: the location maps differ by <= 3e-2 * max|map| (gate above), the
    # read-off is a convex blend of 4 cells x 100 (mm), and the root joint's row is subtracted: 2 * 3e-2 * max|xyz maps| * 100
    bound3 = 2 * 3e-2 * float(np.abs(mf[..., 21:]).max()) * 100
    d3 = np.abs(j3b - j3f)
    print("bf16 vs fp32 joints: %d/21 within one cell, %d/21 on the same pixel; max 3-D diff on those %.3g mm (bound %.3g)"
          % (close.sum(), same.sum(), float(d3[same].max()) if same.any() else -1, bound3))
    # 2-D: heat-maps of random weights are noise-like, so bf16 noise may move an arg-max to another near-maximal cell.  What
    # MUST hold given |bf16 maps - fp32 maps| <= eps everywhere (gate above; the merge and the x8 upsample are convex blends):
    # the fp32 heat-map at the bf16 arg-max is within 2 eps of its own maximum -- for every joint.
    eps = 3e-2 * float(np.abs(mf).max())
    avg_f = oracle.merge_scales(mf, BASELINE_SCALES)[0]
    raw_b = oracle.extract_2d(oracle.merge_scales(mb, BASELINE_SCALES)[0])
    for j in range(21):
        up = oracle.resize(np.ascontiguousarray(avg_f[:, :, j]), 8.0)
        assert up[int(raw_b[j, 0]), int(raw_b[j, 1])] >= up.max() - 2 * eps, j
    # (no "most joints within one cell" floor here: where a heat-map HAS a maximum -- a margin over every other cell that bf16 noise
    # cannot bridge -- the bf16 arg-max must sit in the fp32 cell, for every such joint: test_bf16_margin_conditioned_joints)
    if same[14]:
        assert np.all(d3[same] <= bound3)
    assert float(np.abs(mb - mf).max()) <= 3e-2 * float(np.abs(mf).max())
    hb.close()



def test_bf16_benchmarked_plan_parity(weights, oracle_net, h3):
    """The bf16 plan bench.py times is the ARENA plan (tail GEMM + bone fusion on, S = 3), not the private-buffer one the
    per-layer gate reads back.  (1) its maps equal the private-buffer handle's bit for bit (the fp32 twin of this check is in
    test_conv_stack_every_layer) and sit within 3e-2 of the oracle; (2) over 4 frames incl. non-square ones, for EVERY joint the
    fp32 heat-map at the bf16 arg-max is within 2 eps of its maximum, and the arena handle's joints equal the private-buffer
    handle's; (3) where the heat-maps have a real maximum (planted peaks, utils.py:153-175 semantics) bf16-rounded maps give
    21/21 joints within one heat-map cell."""
    import oracle
    from tests import helpers
    n = _native()
    batch, _, _ = oracle.gen_input_batch(helpers.synth_frame(1234, smooth=True), BASELINE_SCALES)
    ref = oracle_net.forward(batch)
    fused = _handle(BASELINE_SCALES, weights, precision=n.BF16)                              # what bench.py builds
    plain = _handle(BASELINE_SCALES, weights, precision=n.BF16, keep_activations=True)       # what the per-layer gate reads
    names = [L["name"] for L in fused.layers()]
    assert any(">" in x for x in names) and any("bone_length" in x and "deconv" in x for x in names), names   # fused launches present
    assert not any(">" in L["name"] for L in plain.layers())
    mf, mp = fused.forward(batch), plain.forward(batch)
    assert np.array_equal(mf, mp)                                                            # (1)
    assert float(np.abs(mf - ref).max() / np.abs(ref).max()) <= 3e-2
    worst_close = 21
    for k, (H, W) in enumerate([(368, 368), (538, 368), (240, 320), (368, 368)]):           # (2)
        frame = helpers.synth_frame(777 + k, H, W, smooth=True)
        t = T0 + 500 + k / 30
        j2a, j3a = fused.infer(frame, t, t + 0.001)
        j2p, j3p = plain.infer(frame, t, t + 0.001)
        assert np.array_equal(j2a, j2p) and np.array_equal(j3a, j3p), k
        if k == 0:
            h3.reset_filters()
        j2f, j3f = h3.infer(frame, t, t + 0.001)
        mb, m32 = fused.activation("res5c_branch2c"), h3.activation("res5c_branch2c")
        top = float(np.abs(m32).max())
        assert float(np.abs(mb - m32).max()) <= 3e-2 * top, k
        eps = 3e-2 * top
        avg_f = oracle.merge_scales(m32, BASELINE_SCALES)[0]
        raw_b = oracle.extract_2d(oracle.merge_scales(mb, BASELINE_SCALES)[0])
        for j in range(21):
            up = oracle.resize(np.ascontiguousarray(avg_f[:, :, j]), 8.0)
            assert up[int(raw_b[j, 0]), int(raw_b[j, 1])] >= up.max() - 2 * eps, (k, j)
        if k == 0:  # first frame: the filters are the identity, so joints_2d ARE the arg-max positions
            scaler = 368.0 / max(H, W)
            worst_close = min(worst_close, int(np.all(np.abs(j2a - j2f) <= 8.0 / scaler + 1e-9, axis=1).sum()))
    print("bf16 arena plan: worst frame has %d/21 joints within one cell of fp32 (noise heat-maps; informational -- the gate is the "
          "2-eps rule above and test_bf16_margin_conditioned_joints)" % worst_close)
    # (3) planted peaks: the post-processing of bf16-rounded maps against the fp32 maps
    for seed in (5, 6, 7):
        maps = helpers.synth_maps(seed, 3)
        fused.reset_filters()
        a2, a3 = fused.postprocess(_round_bf16(maps), T0, T0 + 0.001)
        fused.reset_filters()
        b2, b3 = fused.postprocess(maps, T0, T0 + 0.001)
        assert np.all(np.abs(a2 - b2) <= 8.0), seed       # 21/21 within one heat-map cell (oracle on the same maps: 0, 1 and 7 px)
    fused.close(), plain.close()


def test_bf16_margin_conditioned_joints(weights):
    """What the bf16 path owes the joints, THROUGH the bf16 net (utils.py:153-219 semantics).  Heat-maps of random weights are noise-like: an
    arg-max may legally jump between near-equal cells (the 2-eps rule of the two tests above), and measured on them NO maximum clears the
    rest of its map by 2 eps (tools/bf16_margin_probe.py: the largest margin of 252 joints is 1.7 eps).  So this test runs weights whose
    heat-maps HAVE maxima -- tests/planted.py: heat-map j peaks ON a blob painted into the frame, over the random net's own noise floor;
    the oracle finds those joints to the pixel, tests/test_planted.py -- through the benchmarked bf16 arena plan and the fp32 plan, 16
    frames (square, the test picture's 538x368, landscape, portrait), every joint:

      * known answer: the fp32 AND the bf16 joints_2d lie on the planted blob, within one heat-map cell (fp32: within one box pixel);
      * map gate: |bf16 maps - fp32 maps| <= eps = 3e-2 * max|fp32 maps| (every frame);
      * the 2-eps rule for every joint; and margin-conditioned exactness: every (frame, joint) whose fp32 maximum beats the best value
        outside its own heat-map cell by more than 2 eps -- merge and x8 upsample are convex blends, bf16 noise cannot bridge that --
        has its bf16 arg-max IN THAT CELL: all of them, and at least 100 such pairs must exist;
      * joints_3d, every joint of every frame: the bf16 handle's read-off against the FP32 location maps read at the same (bf16) pixels
        -- root row included -- within the bound the map gate implies (2 eps x 100 mm, times the read-off's extrapolation weights at the
        borders); and where the pixel AND the root's pixel equal the fp32 path's, against the fp32 path's joints_3d.
    No floor of the kind "most joints within one cell" is left anywhere."""
    import oracle
    from tests import planted
    from tests.test_planted import cell_margin
    n = _native()
    pw = planted.weights()
    hb = _handle(BASELINE_SCALES, pw, precision=n.BF16)         # the arena plan bench.py times
    hf = _handle(BASELINE_SCALES, pw)
    assert any(">" in L["name"] for L in hb.layers())           # fused launches (tail / chain GEMMs) are in this plan
    shapes = [(368, 368), (538, 368), (240, 320), (368, 300)]
    pairs, held, below, rows = 0, 0, 0, []
    worst3, map_err, off_f, off_b = 0.0, 0.0, 0.0, 0.0
    for k in range(16):
        H, W = shapes[k % 4]
        frame, centres = planted.frame(300 + k, H, W)
        want = planted.expected(centres)
        scaler = 368.0 / max(H, W)
        t = T0 + 900 + k
        hb.reset_filters(), hf.reset_filters()                  # first frame of a stream: the filters are the identity
        j2b, j3b = hb.infer(frame, t, t + 0.001)
        mb = hb.activation("res5c_branch2c")
        j2f, j3f = hf.infer(frame, t, t + 0.001)
        mf = hf.activation("res5c_branch2c")
        off_f, off_b = max(off_f, float(np.abs(j2f - want).max()) * scaler), max(off_b, float(np.abs(j2b - want).max()) * scaler)
        assert np.abs(j2f - want).max() <= 1.0 / scaler, (k, np.abs(j2f - want).max())           # the known answer, fp32
        assert np.abs(j2b - want).max() <= 8.0 / scaler, (k, np.abs(j2b - want).max())           # ... and bf16: the same cell or its neighbour
        top = float(np.abs(mf).max())
        eps = 3e-2 * top
        map_err = max(map_err, float(np.abs(mb - mf).max()) / top)
        assert float(np.abs(mb - mf).max()) <= eps, (k, float(np.abs(mb - mf).max()) / top)
        avg_f, avg_b = oracle.merge_scales(mf, BASELINE_SCALES), oracle.merge_scales(mb, BASELINE_SCALES)
        raw_f, raw_b = oracle.extract_2d(avg_f[0]), oracle.extract_2d(avg_b[0])
        for j in range(21):
            up = oracle.resize(np.ascontiguousarray(avg_f[0][:, :, j]), 8.0)
            assert up[int(raw_b[j, 0]), int(raw_b[j, 1])] >= up.max() - 2 * eps, (k, j)          # the 2-eps rule, every joint
            gap = cell_margin(up, raw_f[j])
            same_cell = bool(np.all(raw_b[j] // 8 == raw_f[j] // 8))
            rows.append((k, j, round(gap / eps, 3), same_cell))
            if gap > 2 * eps:
                pairs += 1
                held += same_cell
                assert same_cell, "frame %d joint %d: margin %.2f eps, bf16 arg-max %s left the fp32 cell of %s" % (k, j, gap / eps, raw_b[j], raw_f[j])
            else:
                below += 1
        # joints_3d: the bf16 read-off against the fp32 location maps at the bf16 pixels (first frame: unfiltered positions)
        at_b = oracle.extract_3d(raw_b, avg_f[1], avg_f[2], avg_f[3])
        own = oracle.extract_3d(raw_b, avg_b[1], avg_b[2], avg_b[3])
        assert np.array_equal(own, j3b), k           # the GPU's post-processing of its own maps is exact arithmetic
        # read-off weights: convex inside; a pixel left of 3.5 extrapolates with weights (1 + a, -a), a <= 7/16, per axis
        amp = 1.0 + 2 * (7.0 / 16)
        bound3 = 2 * (eps * 100) * amp * amp
        d3 = float(np.abs(j3b.astype(np.float64) - at_b).max())
        worst3 = max(worst3, d3 / bound3)
        assert d3 <= bound3, (k, d3, bound3)
        same_px = np.all(raw_b == raw_f, axis=1)
        if same_px[14]:
            assert np.all(np.abs(j3b - j3f)[same_px] <= bound3), k
    _log("bf16_margin_pairs.json", {"pairs_with_margin_over_2eps": pairs, "of_them_in_the_fp32_cell": held, "pairs_below": below,
                                    "bf16_map_err_over_max": map_err, "worst_3d_over_bound": worst3, "fp32_offset_from_blob_box_px": off_f,
                                    "bf16_offset_from_blob_box_px": off_b, "rows": rows})
    print("bf16 margin gate: %d of %d (frame, joint) pairs have a margin > 2 eps, all %d in the fp32 cell; bf16 map error %.3g of max (gate 3e-2); "
          "joints vs the planted blobs: fp32 <= %.2f px, bf16 <= %.2f px; 3-D read-off at most %.2f of its bound"
          % (pairs, 16 * 21, held, map_err, off_f, off_b, worst3))
    assert pairs >= 100, "only %d pairs with a real maximum" % pairs
    hb.close(), hf.close()
