"""GPU (MI355X), through the C ABI: pre-processing (a10) and post-processing (a12-a16) against the CPU oracle and against the
reference-recorded fixtures -- bit-exact."""
import json
import os

import numpy as np
import pytest

from tests.gpu_common import BASELINE_SCALES, G, OUT, T0, _EndToEnd, _handle, _log, _native, _round_bf16  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,smooth", [((368, 368), False), ((538, 368), True), ((300, 500), True),
                                          ((720, 1280), True), ((97, 61), False), ((368, 367), False)])
@pytest.mark.parametrize("scales", [BASELINE_SCALES, [1, 0.85, 0.7]])
def test_preprocess_bit_exact(weights, shape, smooth, scales):
    """a10: gen_input_batch on the device == oracle, bit for bit (8-bit fixed-point bilinear + pad + /255-0.4)."""
    import oracle
    from tests import helpers
    frame = helpers.synth_frame(hash(shape) & 0xFFF, shape[0], shape[1], smooth=smooth)
    h = _handle(scales, weights)
    batch, scaler, off = h.preprocess(frame)
    rb, rs, roff = oracle.gen_input_batch(frame, scales)
    h.close()
    assert scaler == rs and off == roff
    assert np.array_equal(batch, rb)


def test_preprocess_fuzz_shapes(h3):
    """a10 over 32 seeded frame shapes (long side 40 .. 1500, aspect ratios up to 6:1, odd sizes): every one bit-exact against the
    oracle, or rejected by both (utils.py:98-103 cannot place a crop whose scaled long side is not 368)."""
    import oracle
    from tests import helpers
    from vnect_amd._native import VnectError
    rng = np.random.RandomState(20240807)
    checked = 0
    for k in range(32):
        long_side = int(rng.randint(40, 1501))
        short = max(8, int(long_side / rng.uniform(1.0, 6.0)))
        H, W = (long_side, short) if k & 1 else (short, long_side)
        frame = helpers.synth_frame(500 + k, H, W, smooth=bool(k & 2))
        try:
            rb, rs, roff = oracle.gen_input_batch(frame, BASELINE_SCALES)
        except Exception:
            with pytest.raises(VnectError):
                h3.preprocess(frame)
            continue
        b, s, off = h3.preprocess(frame)
        assert s == rs and off == roff, (H, W)
        assert np.array_equal(b, rb), (H, W)
        checked += 1
    assert checked >= 24


def test_preprocess_strided_crop(h3):
    """Callers pass crops of a larger frame (run_estimator_ps.py:87): row stride != 3*W."""
    import oracle
    from tests import helpers
    big = helpers.synth_frame(3, 480, 640, smooth=True)
    crop = big[40:400, 100:420]
    b, s, off = h3.preprocess(crop)
    rb, rs, roff = oracle.gen_input_batch(np.ascontiguousarray(crop), BASELINE_SCALES)
    assert s == rs and off == roff and np.array_equal(b, rb)


# ------------------------------------------------------------------------------------------ post-processing
@pytest.mark.parametrize("promo", [0, 1])
def test_postprocess_bit_exact_sequence(weights, promo):
    """a12-a16 on identical maps over 6 frames (filters engaged, irregular dt): bit-exact vs the oracle."""
    import oracle
    from tests import helpers
    h = _handle(BASELINE_SCALES, weights, numpy_promotion=promo)
    ref = oracle.OracleEstimator(scales=BASELINE_SCALES, nep50=bool(promo))
    t = T0
    for k in range(6):
        maps = helpers.synth_maps(300 + k, 3)
        t += 1 / 30 + 0.003 * (k % 3)
        a2, a3 = h.postprocess(maps, t, t + 0.0007, 368 / 538, 58, 0)
        r2, r3 = ref.postprocess(maps, t, t + 0.0007, 368 / 538, 58, 0)
        assert np.array_equal(a2, r2), k
        assert np.array_equal(a3, r3), k
    h.close()


def test_four_scales_whole_path(weights, oracle_net):
    """More than three scales takes the general (8-scale) form of the merge code and a 4-image batch through the conv stack:
    pre-processing and post-processing bit-exact, final maps and joints within the fp32 tolerances."""
    import oracle
    from tests import helpers
    scales = [1.0, 0.9, 0.75, 0.6]
    h = _handle(scales, weights)
    ref = oracle.OracleEstimator(scales=scales, net=oracle_net)
    frame = helpers.synth_frame(321, 400, 310, smooth=True)
    b, sc, off = h.preprocess(frame)
    rb, rs, roff = oracle.gen_input_batch(frame, scales)
    assert np.array_equal(b, rb) and sc == rs and off == roff
    maps, rmaps = h.forward(b), oracle_net.forward(rb)
    assert np.abs(maps - rmaps).max() <= 1e-4 * np.abs(rmaps).max()
    for k in range(3):
        t = T0 + k / 30
        a2, a3 = h.postprocess(rmaps, t, t + 0.0005, sc, off[0], off[1])
        r2, r3 = ref.postprocess(rmaps, t, t + 0.0005, sc, off[0], off[1])
        assert np.array_equal(a2, r2) and np.array_equal(a3, r3), k
    h.close()


def test_six_scales_take_the_unfused_plans(weights, oracle_net):
    """The fused forms of the conv launch (tail GEMM at 92x92, bone features inside the transposed conv) need one tile per
    workgroup; with six scales the 92x92 layers have 794 tiles and the transposed conv 600 items, so the stand-alone layers, the
    dual-output 3x3 launch and the bone kernel run instead (and every tile streams).  Same tolerance against the oracle, and the
    five-scale handle (fused transposed conv, unfused 92x92) agrees too."""
    import oracle
    from tests import helpers
    frame = helpers.synth_frame(2024, 300, 368, smooth=True)
    for scales in ([1.0, 0.95, 0.9, 0.8, 0.7, 0.6], [1.0, 0.9, 0.8, 0.7, 0.6]):
        h = _handle(scales, weights)
        names = [L["name"] for L in h.layers()]
        assert "res2b_branch2b+res2c_branch2b" in names and not any(">" in n for n in names), names
        assert ("res5c_bone_length" in names) == (len(scales) == 6), names
        rb, _, _ = oracle.gen_input_batch(frame, scales)
        b, _, _ = h.preprocess(frame)
        assert np.array_equal(b, rb)
        out, ref = h.forward(b), oracle_net.forward(rb)
        assert float(np.abs(out - ref).max()) <= 1e-4 * float(np.abs(ref).max()), len(scales)
        h.close()


@pytest.mark.parametrize("promo", [0, 1])
def test_postprocess_long_filter_chain(weights, promo):
    """The OneEuro state is a recurrence: 150 frames of moving peaks (a drifting mixture of 5 map sets, so every joint's
    arg-max wanders and the read-off crosses cell borders), irregular frame times, and a timestamp 0.0 ("no timestamp",
    OneEuroFilter.py:65) in the middle -- still bit-exact against the oracle on every frame, no drift."""
    import oracle
    from tests import helpers
    scales = [1.0, 0.7]
    h = _handle(scales, weights, numpy_promotion=promo)
    ref = oracle.OracleEstimator(scales=scales, nep50=bool(promo))
    base = [helpers.synth_maps(700 + k, 2) for k in range(5)]
    t = T0
    for k in range(150):
        w = 0.5 + 0.5 * np.sin(0.13 * k + np.arange(5))
        maps = sum(float(wi) * b for wi, b in zip(w, base)).astype(np.float32)
        t += 1 / 30 + 0.004 * ((k * 7) % 5)
        t2d = 0.0 if k == 70 else t
        a2, a3 = h.postprocess(maps, t2d, t + 0.0004, 1.0, 0, 0)
        r2, r3 = ref.postprocess(maps, t2d, t + 0.0004, 1.0, 0, 0)
        assert np.array_equal(a2, r2) and np.array_equal(a3, r3), k
    h.close()


@pytest.mark.parametrize("case", ["pic_default", "wide_default", "square_baseline", "square_one_scale"])
def test_postprocess_reproduces_reference_recordings(weights, case):
    """The device pre+post-processing reproduces what the reference's own Python returned (fixtures F3)."""
    from tests import helpers
    with np.load(os.path.join(G, "glue_%s.npz" % case)) as z:
        g = {k: z[k] for k in z.files}
    scales = list(g["scales"])
    if case == "pic_default":
        from PIL import Image
        pic = np.asarray(Image.open(os.path.join(G, "test_pic.jpg")).convert("RGB"))[:, :, ::-1].copy()
        frames = [pic] * 3
    elif case == "wide_default":
        frames = [helpers.synth_frame(31 + k, 300, 500, smooth=True) for k in range(3)]
    elif case == "square_baseline":
        frames = [helpers.synth_frame(1234 + k) for k in range(4)]
    else:
        frames = [helpers.synth_frame(77, smooth=True)]
    h = _handle(scales, weights, numpy_promotion=1)  # fixtures were recorded under numpy 2.x
    for k, frame in enumerate(frames):
        batch, scaler, (ox, oy) = h.preprocess(frame)
        assert [scaler, ox, oy] == list(g["meta"][k])
        assert np.array_equal(batch.astype(np.float64).sum(axis=(1, 2, 3)), g["batch_sum"][k])
        assert np.array_equal(batch[:, ::37, ::41, :], g["batch_probe"][k])
        maps = helpers.synth_maps(int(g["map_seed"]) + k, len(scales))
        j2, j3 = h.postprocess(maps, g["t2d"][k], g["t3d"][k], scaler, ox, oy)
        assert np.array_equal(j2, g["joints_2d"][k]), k
        assert np.array_equal(j3, g["joints_3d"][k]), k
    h.close()


def test_argmax_ties_and_planted_peaks(weights):
    h = _handle([1.0], weights)
    flat = np.zeros((1, 46, 46, 84), np.float32)
    j2, j3 = h.postprocess(flat, T0, T0)
    assert np.all(j2 == 0) and np.all(j3 == 0)  # first maximum in row-major order
    hm = np.zeros((1, 46, 46, 84), np.float32)
    yy, xx = np.mgrid[0:46, 0:46]
    cells = [(3 + 2 * j, 40 - j) for j in range(21)]
    for j, (cy, cx) in enumerate(cells):
        hm[0, :, :, j] = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 1.5 ** 2))
    h.reset_filters()
    j2, _ = h.postprocess(hm, T0, T0)
    for j, (cy, cx) in enumerate(cells):
        assert abs(j2[j, 0] - (cy * 8 + 3.5)) <= 0.5 and abs(j2[j, 1] - (cx * 8 + 3.5)) <= 0.5
    h.close()
