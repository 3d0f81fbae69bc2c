"""GPU (MI355X), through the C ABI: the whole __call__ (a9-a17) against the oracle with the every-frame, every-joint gate
(tests/gpu_common.py: _EndToEnd), frames in flight, the tracking loop, planted known answers."""
import json
import os

import numpy as np
import pytest

from tests.gpu_common import BASELINE_SCALES, G, OUT, T0, _EndToEnd, _handle, _log, _native, _round_bf16  # noqa: F401

pytestmark = pytest.mark.gpu


def test_end_to_end_vs_oracle(h3, oracle_net):
    """Whole __call__ over 4 frames against the oracle: see _EndToEnd (every frame, every joint)."""
    from tests import helpers
    h3.reset_filters()
    e2e = _EndToEnd(BASELINE_SCALES, oracle_net)
    for k in range(4):
        frame = helpers.synth_frame(1234 + k, smooth=True)
        t = T0 + k / 30
        j2, j3 = h3.infer(frame, t, t + 0.001)
        e2e.check(frame, t, t + 0.001, j2, j3, h3.activation("res5c_branch2c"), k)
    print("legal arg-max ties: %d, worst 3-D excess over tolerance: %.3g" % (e2e.ties, e2e.worst3))


def test_end_to_end_nonsquare_frames(h3, oracle_net):
    """Whole __call__ on frames that are not 368x368 (the size of pic/test_pic.jpg, a landscape VGA-like crop, a small portrait
    one): squarify scaler and centring offsets enter the un-mapping (estimator.py:137-139).  Same gate as the square case."""
    from tests import helpers
    h3.reset_filters()
    e2e = _EndToEnd(BASELINE_SCALES, oracle_net)
    for k, (H, W) in enumerate([(538, 368), (240, 320), (200, 120), (538, 368)]):
        frame = helpers.synth_frame(4321 + k, H, W, smooth=True)
        t = T0 + 100 + k / 30
        j2, j3 = h3.infer(frame, t, t + 0.001)
        e2e.check(frame, t, t + 0.001, j2, j3, h3.activation("res5c_branch2c"), (H, W))


@pytest.mark.parametrize("scales", [[1, 0.85, 0.7], [1.0, 0.7], [1.0]])
def test_end_to_end_reference_default_scales(weights, oracle_net, scales):
    """Whole __call__ at the reference's own pyramid (estimator.py:32: [1, 0.85, 0.7]) and at the two shorter ones its comment
    suggests "for faster loops" ([1, 0.7], [1]), on frames of the test picture's size: the every-frame, every-joint gate."""
    from tests import helpers
    h = _handle(scales, weights)
    e2e = _EndToEnd([float(s) for s in scales], oracle_net)
    for k in range(3):
        frame = helpers.synth_frame(3100 + k, 538, 368, smooth=True)
        t = T0 + 200 + k / 30 + 0.002 * k
        j2, j3 = h.infer(frame, t, t + 0.0013)
        e2e.check(frame, t, t + 0.0013, j2, j3, h.activation("res5c_branch2c"), (scales, k))
    h.close()


@pytest.mark.parametrize("lanes,graph", [(1, True), (2, True), (2, False), (3, True)])
def test_pipelined_submit_collect_equals_sequential(weights, lanes, graph):
    """Two frames in flight (submit k+1 before collecting k) return exactly what one-at-a-time inference returns: on one
    lane (same stream), and on two lanes (lanes=2: the frames overlap on two streams / activation arenas and only the
    joints kernels -- the OneEuro filter chain -- are ordered by an event).  Frames of different sizes alternate, so each
    lane keeps its own crop geometry; 12 frames exercise both lanes and the 4-deep result ring several times."""
    from tests import helpers
    shapes = [(368, 368), (300, 420), (368, 368), (410, 260)]
    frames = [helpers.synth_frame(500 + k, *shapes[k], smooth=True) for k in range(4)]
    a = _handle(BASELINE_SCALES, weights, lanes=lanes, use_graph=graph)
    b = _handle(BASELINE_SCALES, weights, use_graph=False)
    for k, f in enumerate(frames):
        a.upload_frame(k, f)
        b.upload_frame(k, f)
    n = 12
    seq = [b.infer_resident(k % 4, T0 + k / 30, T0 + k / 30 + 0.001) for k in range(n)]
    depth = max(lanes, 2)  # frames kept in flight
    got = []
    for k in range(n):
        if k >= depth:
            got.append(a.collect())
        a.submit_resident(k % 4, T0 + k / 30, T0 + k / 30 + 0.001)
    for _ in range(depth):
        got.append(a.collect())
    with pytest.raises(_native().VnectError):
        a.collect()  # nothing left in flight
    for k, ((g2, g3), (s2, s3)) in enumerate(zip(got, seq)):
        assert np.array_equal(g2, s2) and np.array_equal(g3, s3), k  # bit for bit
    # back to one at a time on the same handle: the filter chain continues across the mode change
    k = n
    g2, g3 = a.infer_resident(1, T0 + k / 30, T0 + k / 30 + 0.001)
    s2, s3 = b.infer_resident(1, T0 + k / 30, T0 + k / 30 + 0.001)
    assert np.array_equal(g2, s2) and np.array_equal(g3, s3)
    a.close(), b.close()


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp32_split"])
def test_soak_three_lanes_deterministic(weights, prec):
    """Race detector: 3 000 frames of one stream, three in flight on three lanes, twice, and once frame by frame -- all
    three result sequences must be identical bit for bit (the K-group hand-off through LDS, the lane events, the result
    ring and the arena sharing all have to be right every single time for that)."""
    from tests import helpers
    frames = [helpers.synth_frame(40 + k, smooth=(k % 2 == 0)) for k in range(8)]
    h = _handle(BASELINE_SCALES, weights, lanes=3, num_frame_slots=8,
                precision={"bf16": _native().BF16, "fp32_split": _native().FP32_SPLIT}.get(prec, _native().FP32))
    for k, f in enumerate(frames):
        h.upload_frame(k, f)
    n = 3000

    def run(depth):
        h.reset_filters()
        out = np.empty((n, 21, 5), np.float64)
        got = 0
        for k in range(n):
            if k >= depth:
                j2, j3 = h.collect()
                out[got, :, :2], out[got, :, 2:] = j2, j3
                got += 1
            h.submit_resident((k * 3) % 8, T0 + k / 30, T0 + k / 30 + 0.0005)
        while got < n:
            j2, j3 = h.collect()
            out[got, :, :2], out[got, :, 2:] = j2, j3
            got += 1
        return out

    a, b, c = run(3), run(3), run(1)
    h.close()
    assert np.all(np.isfinite(a))
    assert np.array_equal(a, b), "two pipelined runs differ at frame %d" % int(np.argmax(np.any(a != b, axis=(1, 2))))
    assert np.array_equal(a, c), "pipelined and frame-by-frame runs differ at frame %d" % int(np.argmax(np.any(a != c, axis=(1, 2))))


def test_estimator_submit_collect_two_lanes(weights):
    """The facade's additive pipelined API (submit / collect, lanes=2) against its own frame-by-frame __call__, with frames
    uploaded from host memory each time (an upload waits only for the inference that still reads its slot)."""
    from tests import helpers
    from vnect_amd import VNectEstimator
    frames = [helpers.synth_frame(900 + k, 368 - 11 * (k % 3), 300 + 17 * (k % 4), smooth=True) for k in range(9)]
    one = VNectEstimator(scales=BASELINE_SCALES, weights=weights, verbose=False)
    two = VNectEstimator(scales=BASELINE_SCALES, weights=weights, verbose=False, lanes=2)
    want = [one(f, timestamp=T0 + k / 30) for k, f in enumerate(frames)]
    got = []
    two.submit(frames[0], timestamp=T0)
    for k in range(1, len(frames)):
        two.submit(frames[k], timestamp=T0 + k / 30)
        got.append(two.collect())
    got.append(two.collect())
    for k, ((g2, g3), (w2, w3)) in enumerate(zip(got, want)):
        assert np.array_equal(g2, w2) and np.array_equal(g3, w3), k
    with pytest.raises(_native().VnectError):  # a third frame in flight is refused, state untouched
        two.submit(frames[0], timestamp=T0 + 1)
        two.submit(frames[1], timestamp=T0 + 2)
        two.submit(frames[2], timestamp=T0 + 3)
    one.close(), two.close()


def test_errors_mirror_reference(weights):
    from vnect_amd import VNectEstimator
    est = VNectEstimator(scales=[1.0], weights=weights, verbose=False)
    frame = np.zeros((368, 368, 3), np.uint8)
    est(frame, timestamp=5.0)
    with pytest.raises(ZeroDivisionError):   # OneEuroFilter.py:66
        est(frame, timestamp=5.0)
    with pytest.raises(ValueError):
        est(np.zeros((368, 368), np.uint8))
    j2, j3 = est(frame, timestamp=6.0)
    assert j2.shape == (21, 2) and j2.dtype == np.float64 and j3.shape == (21, 3) and j3.dtype == np.float32
    est.scales = [1.0, 0.7]   # assignable like the reference attribute
    j2, j3 = est(frame, timestamp=7.0)
    assert np.all(np.isfinite(j2))
    est.close()


def test_tracking_loop_variable_crops(weights, oracle_net):
    """run_estimator_ps.py:80-109 headless: the crop changes every frame, so squarify/resize tables are rebuilt per
    call; every frame and every joint is checked against the oracle fed the same crop (_EndToEnd)."""
    from vnect_amd import VNectEstimator, runner
    scales = [1.0, 0.8, 0.6]
    est = VNectEstimator(scales=scales, weights=weights, verbose=False)
    e2e = _EndToEnd(scales, oracle_net)
    frames = list(runner.synthetic_stream(3, 4, 480, 640))
    rect, sizes = [40, 30, 500, 400], set()
    for k, frame in enumerate(frames):
        x, y, w, h = rect
        crop = frame[y:y + h, x:x + w, :]
        sizes.add(crop.shape)
        t = T0 + k / 30
        j2, j3 = est(crop, timestamp=(t, t + 0.001))
        e2e.check(crop, t, t + 0.001, j2, j3, est.handle.activation("res5c_branch2c"), k)
        j2[:, 0] += y
        j2[:, 1] += x
        rect = runner.bbox_update(j2, 640, 480)
        if rect[2] < 8 or rect[3] < 8:
            rect = [0, 0, 640, 480]
    assert len(sizes) >= 2   # the loop really exercised more than one crop geometry
    est.close()


def test_tracking_loop_follows_planted_blobs():
    """The tracking loop with a KNOWN answer on the GPU (tests/planted.py; CPU twin with the oracle: tests/test_planted.py): a 640 x 480 video of
    three drifting blobs through runner.track (run_estimator_ps.py:80-109) -- whole frame first, then crops by the box rule, every crop a
    different size, squarified and resized on the device.  fp32: the loop is the ORACLE's loop joint for joint (joints_2d array_equal in every
    frame -- planted peaks leave no ties --, hence the same crops; joints_3d within the fp32 tolerance); fp32 and bf16: every joint within two
    heat-map cells of its blob, every crop within 2.5 cells + 8 pixels of the box rule applied to the true positions."""
    import oracle
    from tests import planted
    from tests.test_planted import moving_person
    from vnect_amd import VNectEstimator, runner
    H, W, n = 480, 640, 8
    pw = planted.weights()
    net = oracle.Oracle(pw)
    frames = [planted.scene(H, W, moving_person(k), sigma=10.0, seed=k) for k in range(n)]
    stamps = [T0 + 50 + i / 30 for i in range(n)]

    class OracleEst:
        def __init__(self):
            self.o = oracle.OracleEstimator(scales=BASELINE_SCALES, net=net)

        def __call__(self, img, timestamp=None):
            return self.o(np.ascontiguousarray(img), timestamp, timestamp)

    ref = list(runner.track(OracleEst(), frames, timestamps=stamps))
    for prec in ("fp32", "bf16"):
        est = VNectEstimator(scales=BASELINE_SCALES, weights=pw, precision=prec, verbose=False)
        prev_ideal, worst = None, 0.0
        for k, (j2, j3, rect) in enumerate(runner.track(est, frames, timestamps=stamps)):
            want = np.array([moving_person(k)[j % 3][:2] for j in range(21)], np.float64)
            cell = 8.0 / (368.0 / max(rect[2], rect[3]))
            worst = max(worst, float(np.abs(j2 - want).max()) / cell)
            assert np.abs(j2 - want).max() <= max(2.0 * cell, 14.0), (prec, k, rect, float(np.abs(j2 - want).max()))
            if prev_ideal is not None:
                assert np.abs(np.array(rect) - np.array(prev_ideal)).max() <= 2.5 * prev_cell + 8, (prec, k, rect, prev_ideal)
            prev_ideal, prev_cell = runner.bbox_update(want, W, H), cell
            if prec == "fp32":
                r2, r3, rrect = ref[k]
                assert rect == rrect and np.array_equal(j2, r2), (k, rect, rrect)
                assert np.all(np.abs(j3 - r3) <= 0.05 + 1e-4 * np.abs(r3)), k
        print("%s: tracked %d frames, joints at most %.2f heat-map cells from their blobs" % (prec, n, worst))
        est.close()


def test_planted_joints_through_every_way_of_running_a_frame():
    """The geometric known answer of tests/planted.py (joint j ON the bright blob of colour j % 3) through every way the library runs a frame
    -- each of them bit-equal to the synchronous call elsewhere in this file; here they must also be RIGHT: frames in flight on three lanes,
    three videos on one handle, the split-product path, a pyramid sharded over three rank handles (each builds and runs ONE scale; the maps
    are stacked in rank order as the exchange delivers them -- a swapped or stale slot would pull the merged peaks off the blobs), and one
    scale alone.  Tolerance: one box pixel (the oracle's own: tests/test_planted.py)."""
    from tests import planted
    n = _native()
    pw = planted.weights()
    frames = [planted.frame(700 + k, H, W) for k, (H, W) in enumerate([(368, 368), (538, 368), (240, 320), (368, 368)])]

    def on_blobs(j2, k, tol=1.0):
        frame, centres = frames[k]
        scaler = 368.0 / max(frame.shape[:2])
        d = float(np.abs(j2 - planted.expected(centres)).max())
        assert d <= tol / scaler, (k, d, scaler)

    # frames in flight on three lanes (a new stream per frame would reset the filters; one stream: the blobs jump between frames, so
    # only the first frame is unfiltered -- use one handle per check instead and submit the SAME frame three times: the filters settle on it)
    h = _handle(BASELINE_SCALES, pw, lanes=3)
    for k in range(3):
        h.upload_frame(k, frames[0][0])
        h.submit_resident(k, T0 + k / 30, T0 + k / 30 + 0.001)
    for k in range(3):
        on_blobs(h.collect()[0], 0)
    # three videos on one handle: stream s sees frame s
    h.reset_filters()
    for s_ in range(3):
        h.upload_frame(s_, frames[s_][0])
        h.submit_stream(s_, s_, T0 + 10 + s_, T0 + 10 + s_ + 0.001)
    for s_ in range(3):
        st, j2, j3 = h.collect_stream()
        on_blobs(j2, st)
    h.close()
    # the split-product path and a single scale
    for kw, scales in ((dict(precision=n.FP32_SPLIT), BASELINE_SCALES), (dict(), [1.0]), (dict(precision=n.BF16), [1.0, 0.8])):
        h = _handle(scales, pw, **kw)
        for k in range(len(frames)):
            h.reset_filters()
            on_blobs(h.infer(frames[k][0], T0 + 20 + k, T0 + 20 + k + 0.001)[0], k, tol=8.0 if kw.get("precision") == n.BF16 else 1.0)
        h.close()
    # pyramid sharded over three rank handles: rank r pre-processes and runs scale r; the stack in rank order is what the exchange delivers
    ranks = [n.Handle(BASELINE_SCALES, pyramid=(r, 3)) for r in range(3)]
    for hh in ranks:
        hh.set_weights(pw)
        hh.finalize()
    for k in range(len(frames)):
        maps = []
        for hh in ranks:
            b, scaler, (ox, oy) = hh.preprocess(frames[k][0])
            maps.append(hh.forward(b)[0])
        ranks[0].reset_filters()
        j2, j3 = ranks[0].postprocess(np.stack(maps), T0 + 30 + k, T0 + 30 + k + 0.001, scaler, ox, oy)
        on_blobs(j2, k)
        if k == 1:  # the wrong slot order is NOT right: the test can see what it claims to see
            ranks[0].reset_filters()
            w2, _ = ranks[0].postprocess(np.stack([maps[1], maps[0], maps[2]]), T0 + 40, T0 + 40.001, scaler, ox, oy)
            frame, centres = frames[k]
            assert float(np.abs(w2 - planted.expected(centres)).max()) > 8.0 / (368.0 / max(frame.shape[:2]))
    for hh in ranks:
        hh.close()


@pytest.mark.parametrize("scales", [[1.0], None])
def test_run_pic_on_the_reference_picture(weights, oracle_net, scales):
    """BASELINE.json configs[0] on the GPU: the flow of /root/reference/run_pic.py:18-30 -- read pic/test_pic.jpg (committed as
    tests/golden/test_pic.jpg), take the full-frame rectangle (the no-detection fallback, src/hog_box.py:28-29), estimate, add the crop
    origin -- through `runner.run_pic(VNectEstimator(...))` at one scale and at the reference's default pyramid, against the oracle with
    the every-frame, every-joint gate.  A second call with a real rectangle makes the crop offsets count (estimator.py:137-139 +
    run_pic.py:22-24)."""
    from vnect_amd import VNectEstimator, runner
    img = runner.load_bgr(os.path.join(G, "test_pic.jpg"))
    est = VNectEstimator(scales=scales, weights=weights, verbose=False)

    class Tap:  # what the estimator itself returned for the crop, before run_pic shifted it
        def __call__(self, crop, timestamp=None):
            j2, j3 = est(crop, timestamp=timestamp)
            self.raw = (j2.copy(), j3.copy())
            return j2, j3

    tap = Tap()
    e2e = _EndToEnd(est.scales, oracle_net)
    H, W = img.shape[:2]
    for k, rect in enumerate([None, [40, 60, 300, 420]]):
        t = (T0 + k / 30, T0 + k / 30 + 0.001)
        j2, j3, used = runner.run_pic(tap, img, rect=rect, timestamp=t)
        x, y, w, h = used
        assert used == (rect if rect is not None else [0, 0, W, H])
        assert j2.shape == (21, 2) and j2.dtype == np.float64 and j3.shape == (21, 3) and j3.dtype == np.float32
        e2e.check(img[y:y + h, x:x + w, :], t[0], t[1], tap.raw[0], tap.raw[1], est.handle.activation("res5c_branch2c"), (scales, rect))
        assert np.array_equal(j2, tap.raw[0] + np.array([y, x], np.float64)) and np.array_equal(j3, tap.raw[1])   # run_pic.py:22-24
    est.close()
