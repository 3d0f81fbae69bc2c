"""CPU: a geometric known answer for the oracle's whole path -- planted peaks through the network (tests/planted.py).

The oracle's conv stack has no reference-held vectors (TF1 cannot run, no weights ship): it is cross-checked against an independent
float64 restatement (tests/test_oracle_net.py).  This test adds something neither restatement can fake: weights built so that heat-map j
must peak ON a blob painted into the frame, to the pixel, after the whole graph of src/vnect_model.py:27-217, the multi-scale merge
(src/estimator.py:105-129) and extract_2d_joints (src/utils.py:153-175).  The GPU twin (fp32 and bf16, with the margin-conditioned
bf16 gate) is tests/test_gpu_bf16.py::test_bf16_margin_conditioned_joints.
"""
import numpy as np
import pytest

BASELINE_SCALES = [1.0, 0.8, 0.6]


def cell_margin(up, rc):
    """The maximum of the x8-upsampled heat-map minus its best value OUTSIDE the 8 x 8 block of pixels (one heat-map cell) that holds it."""
    r0, c0 = (int(rc[0]) // 8) * 8, (int(rc[1]) // 8) * 8
    rest = up.copy()
    rest[r0:r0 + 8, c0:c0 + 8] = -np.inf
    return float(up[int(rc[0]), int(rc[1])] - rest.max())


@pytest.fixture(scope="module")
def planted_net():
    import oracle
    from tests import planted
    return oracle.Oracle(planted.weights())


@pytest.mark.parametrize("seed,shape", [(1, (368, 368)), (2, (538, 368)), (3, (240, 320))])
def test_oracle_finds_the_planted_joints(planted_net, seed, shape):
    import oracle
    from tests import planted
    H, W = shape
    frame, centres = planted.frame(seed, H, W)
    want = planted.expected(centres)
    est = oracle.OracleEstimator(scales=BASELINE_SCALES, net=planted_net)
    j2, j3 = est(frame, 1.0, 1.0)                       # first frame of a stream: the filters are the identity
    scaler = 368.0 / max(H, W)
    assert np.abs(j2 - want).max() <= 1.0 / scaler, (np.abs(j2 - want).max(), scaler)    # ON the blob: within one box pixel
    assert np.all(np.isfinite(j3))
    # and these are REAL maxima: most of them clear every other cell by more than 2 eps, eps = 3e-2 * max|maps| (the bf16 map gate)
    batch, _, _ = oracle.gen_input_batch(frame, BASELINE_SCALES)
    maps = planted_net.forward(batch)
    eps = 3e-2 * float(np.abs(maps).max())
    avg = oracle.merge_scales(maps, BASELINE_SCALES)[0]
    raw = oracle.extract_2d(avg)
    m = [cell_margin(oracle.resize(np.ascontiguousarray(avg[:, :, j]), 8.0), raw[j]) / eps for j in range(21)]
    assert sum(x > 2.0 for x in m) >= 12, m


def test_planted_weights_keep_the_schema_and_isolate_the_pass_channels():
    from tests import planted
    from vnect_amd.weights import check_schema, synthetic_weights
    w, base = planted.weights(), synthetic_weights()
    check_schema(w)
    for name, a in w.items():
        scope, leaf = name.split("/")
        if leaf == "weights" and scope not in ("conv1", "res5c_branch2b"):   # (res5c_branch2b fans the 3 colours out to 21 joint channels)
            # no random output reads a pass channel, no pass output reads a random channel
            assert not np.any(a[:, :, :planted.P, planted.P:]) and not np.any(a[:, :, planted.P:, :planted.P]), name
    k = w["res5c_branch2b/weights"]
    assert not np.any(k[:, :, planted.P:, :21]) and not np.any(k[:, :, :planted.P, 21:]) and np.count_nonzero(k[:, :, :planted.P, :21]) == 21
    # the random part is the seeded synthetic net
    k, b = w["res4c_branch2b/weights"], base["res4c_branch2b/weights"]
    assert np.array_equal(k[:, :, planted.P:, planted.P:], b[:, :, planted.P:, planted.P:])


def moving_person(k):
    """three blobs (colour 0, 1, 2: the 'person' every joint j sits on by j % 3) drifting right and down by (5, 2) pixels per frame"""
    return [(150 + 2 * k, 250 + 5 * k, 0, 255), (300 + 2 * k, 300 + 5 * k, 1, 255), (330 + 2 * k, 220 + 5 * k, 2, 255)]


def test_tracking_loop_locks_onto_the_planted_person(planted_net):
    """The caller loop of run_estimator_ps.py:80-109 (vnect_amd.runner.track) with a known answer: a 640 x 480 video of three drifting blobs.
    From the whole frame the loop must find them, crop around them by the box rule (:96-107) and follow them: every joint within two
    heat-map cells of its blob in every frame, every crop within 2.5 cells (of the crop it was measured in) + 8 pixels of the box rule applied to the true positions.  (The oracle
    behind the estimator surface; the GPU twin -- fp32 equal to this loop joint for joint, and bf16 -- is
    tests/test_gpu_end_to_end.py::test_tracking_loop_follows_planted_blobs.)"""
    import oracle
    from tests import planted
    from vnect_amd import runner
    H, W, n = 480, 640, 5

    class OracleEst:
        def __init__(self):
            self.o = oracle.OracleEstimator(scales=BASELINE_SCALES, net=planted_net)

        def __call__(self, img, timestamp=None):
            return self.o(np.ascontiguousarray(img), timestamp, timestamp)

    frames = [planted.scene(H, W, moving_person(k), sigma=10.0, seed=k) for k in range(n)]
    prev_ideal = None
    for k, (j2, j3, rect) in enumerate(runner.track(OracleEst(), frames, timestamps=[10 + i / 30 for i in range(n)])):
        want = np.array([moving_person(k)[j % 3][:2] for j in range(21)], np.float64)
        cell = 8.0 / (368.0 / max(rect[2], rect[3]))                      # one heat-map cell in frame pixels for this crop
        assert np.abs(j2 - want).max() <= max(2.0 * cell, 14.0), (k, rect, float(np.abs(j2 - want).max()))
        if prev_ideal is not None:    # (a joint error of e pixels moves a box edge by up to 1.8 e: the rule adds 80 % of the width)
            assert np.abs(np.array(rect) - np.array(prev_ideal)).max() <= 2.5 * prev_cell + 8, (k, rect, prev_ideal)
        else:
            assert rect == [0, 0, W, H]
        prev_ideal, prev_cell = runner.bbox_update(want, W, H), cell
