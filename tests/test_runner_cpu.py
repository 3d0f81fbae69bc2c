"""CPU: host logic of the headless runner (run_estimator_ps.py:96-107 bounding-box update, run_pic flow)."""
import os

import numpy as np

from vnect_amd import runner

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class FakeEstimator:
    """Returns fixed joints in crop coordinates and records the crops it was given."""

    def __init__(self, j2):
        self.j2, self.crops = j2, []

    def __call__(self, img, timestamp=None):
        self.crops.append(img.shape)
        return self.j2.copy(), np.zeros((21, 3), np.float32)

    def reset(self):
        self.resets = getattr(self, "resets", 0) + 1


def test_bbox_update_matches_reference_arithmetic():
    j2 = np.zeros((21, 2))
    j2[:, 0] = np.linspace(100.5, 300.25, 21)   # rows
    j2[:, 1] = np.linspace(200.0, 260.75, 21)   # cols
    x, y, w, h = runner.bbox_update(j2, 640, 480)
    # run_estimator_ps.py:100-107 by hand
    bx, by = 0.8 * (260.75 - 200.0 + 1), 0.2 * (300.25 - 100.5 + 1)
    assert (x, y) == (int(200.0 - bx / 2), int(100.5 - by / 2))
    assert (w, h) == (int(min(60.75 + bx, 640 - x)), int(min(199.75 + by, 480 - y)))
    # clamping at the frame border
    j2[:, 1] += 400
    x, y, w, h = runner.bbox_update(j2, 640, 480)
    assert x + w <= 640 and y + h <= 480 and x >= 0


def test_run_pic_flow_and_offsets():
    img = runner.load_bgr(os.path.join(G, "test_pic.jpg"))
    assert img.shape == (538, 368, 3) and img.dtype == np.uint8
    assert runner.full_frame_rect(img) == [0, 0, 368, 538]
    j2 = np.tile(np.array([[10.0, 20.0]]), (21, 1))
    est = FakeEstimator(j2)
    out2, _, rect = runner.run_pic(est, img, rect=[5, 7, 100, 200])
    assert est.crops == [(200, 100, 3)] and rect == [5, 7, 100, 200]
    assert np.all(out2[:, 0] == 17.0) and np.all(out2[:, 1] == 25.0)   # += y, += x  (run_pic.py:29-30)


def test_track_loop_updates_crop():
    frames = list(runner.synthetic_stream(0, 3, 240, 320))
    assert frames[0].shape == (240, 320, 3) and not np.array_equal(frames[0], frames[1])
    j2 = np.zeros((21, 2))
    j2[:, 0] = np.linspace(20, 120, 21)
    j2[:, 1] = np.linspace(30, 90, 21)
    est = FakeEstimator(j2)
    out = list(runner.track(est, frames, timestamps=[1.0, 2.0, 3.0]))
    assert len(out) == 3 and out[0][2] == [0, 0, 320, 240]
    assert out[1][2] == runner.bbox_update(out[0][0], 320, 240)     # frame k+1 is cropped by frame k's joints
    assert est.crops[1] == (out[1][2][3], out[1][2][2], 3)
    # different streams are different videos, same stream is reproducible
    a = next(runner.synthetic_stream(1, 1)); b = next(runner.synthetic_stream(2, 1)); c = next(runner.synthetic_stream(1, 1))
    assert not np.array_equal(a, b) and np.array_equal(a, c)


def test_init_box_probes_the_whole_frame_then_resets():
    """the box initialiser = full-frame pass (the reference's no-detection rectangle, hog_box.py:28-29)
    + the loop's box arithmetic; the probe must not leave filter state behind."""
    frame = next(runner.synthetic_stream(3, 1, 240, 320))
    j2 = np.zeros((21, 2))
    j2[:, 0] = np.linspace(40, 200, 21)
    j2[:, 1] = np.linspace(100, 180, 21)
    est = FakeEstimator(j2)
    rect = runner.init_box(est, frame, timestamp=5.0)
    assert est.crops == [(240, 320, 3)] and est.resets == 1
    assert rect == runner.bbox_update(j2, 320, 240)
    # all joints on one pixel -> degenerate box -> whole frame
    est2 = FakeEstimator(np.full((21, 2), 7.0))
    x, y, w, h = runner.bbox_update(est2.j2, 320, 240)
    assert runner.init_box(est2, frame) == ([x, y, w, h] if w >= 1 and h >= 1 else [0, 0, 320, 240])


def test_render_2d_and_3d_to_files(tmp_path):
    """SURVEY 8f rank 4 (rendering): limbs and rectangle land where the joints are; files are written and re-readable."""
    from vnect_amd import render
    from vnect_amd.estimator import VNectEstimator
    img = np.full((240, 320, 3), 255, np.uint8)
    j2 = np.zeros((21, 2))
    j2[:, 0] = np.linspace(40, 200, 21)     # rows
    j2[:, 1] = np.linspace(100, 180, 21)    # cols
    out = render.draw_limbs_2d(img, j2, VNectEstimator.joint_parents, [90, 30, 100, 180])
    assert out.shape == img.shape and out.dtype == np.uint8 and np.all(img == 255)   # the input is not drawn on
    r, c = int(round((j2[2, 0] + j2[1, 0]) / 2)), int(round((j2[2, 1] + j2[1, 1]) / 2))
    assert tuple(out[r, c]) == render.LIMB_BGR                     # middle of limb 2 -> 1 (BGR)
    assert tuple(out[30, 140]) == render.RECT_BGR and tuple(out[5, 5]) == (255, 255, 255)
    j3 = np.zeros((21, 3), np.float32)
    j3[:, 0] = np.linspace(-300, 300, 21)
    j3[:, 1] = np.linspace(-400, 400, 21)
    v = render.draw_limbs_3d(j3, VNectEstimator.joint_parents)
    assert v.shape == (400, 400, 3) and (v != 255).any()
    p = tmp_path / "pose.png"
    render.save(str(p), out)
    assert np.array_equal(runner.load_bgr(str(p)), out)            # PNG round trip
