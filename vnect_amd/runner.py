"""Headless counterparts of the reference's runner scripts (SURVEY 8 f-row 1, BASELINE.json configs[0]).

* ``run_pic``   -- the flow of ``run_pic.py:18-30`` without cv2 / HOG / GUI: decode the picture, take the
  full-frame rectangle (the reference's no-detection fallback, ``src/hog_box.py:28-29``), estimate, shift the
  2-D joints back by the crop origin.
* ``track``     -- the frame loop of ``run_estimator_ps.py:80-109``: crop -> estimator -> bounding-box update from
  the joints, for any iterable of frames (synthetic streams on the GPU box; there is no camera / ffmpeg).
* ``init_box``  -- the person-box initialiser: the reference's HOG detector (``src/hog_box.py``) replaced by one pass of
  the network over the whole frame and the loop's own box arithmetic.
The capture, drawing and 3-D plotting of the reference stay out of scope.
"""
import numpy as np


def load_bgr(path):
    """cv2.imread replacement: uint8 BGR (H, W, 3) via PIL (cv2 is not available on either box)."""
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])


def full_frame_rect(img):
    """HOGBox's fallback rectangle when no person is detected (src/hog_box.py:28-29): (x, y, w, h)."""
    h, w = img.shape[:2]
    return [0, 0, w, h]


def run_pic(estimator, img, rect=None, timestamp=None):
    """run_pic.py:18-30: returns (joints_2d in full-image [row, col], joints_3d, rect)."""
    x, y, w, h = rect if rect is not None else full_frame_rect(img)
    img_cropped = img[y: y + h, x: x + w, :]
    joints_2d, joints_3d = estimator(img_cropped, timestamp=timestamp)
    joints_2d[:, 0] += y
    joints_2d[:, 1] += x
    return joints_2d, joints_3d, [x, y, w, h]


def bbox_update(joints_2d, W_img, H_img):
    """The tracking loop's next crop rectangle (x, y, w, h) from the 2-D joints in frame coordinates -- the rule of
    run_estimator_ps.py:96-107: the joints' bounding box grown by 80 % of (its width + 1) and 20 % of (its height + 1), half of the
    margin before the box, the whole margin added to its extent, clamped to the frame; integer truncation as there."""
    lo = joints_2d.min(axis=0)                 # [row, col] = [y, x]
    span = joints_2d.max(axis=0) - lo
    rect = {}
    for axis, grow, limit in ((1, 0.8, W_img), (0, 0.2, H_img)):
        margin = grow * (span[axis] + 1)
        origin = max(int(lo[axis] - margin / 2), 0)
        rect[axis] = (origin, int(min(span[axis] + margin, limit - origin)))
    (x, w), (y, h) = rect[1], rect[0]
    return [x, y, w, h]


def init_box(estimator, frame, timestamp=None):
    """Person-box initialiser without cv2's HOG detector (SURVEY 8 f-row 3; the reference: src/hog_box.py:25-58).

    The reference asks a HOG people detector for a rectangle and falls back to the whole frame when it finds nobody
    (hog_box.py:28-29).  Here the network itself proposes the box: one pass over the whole frame (that fallback
    rectangle), then the tracking loop's own box arithmetic on the joints it found (run_estimator_ps.py:96-107).  The
    pass is a probe: the estimator's filters are reset afterwards, so the first tracked frame is still an unfiltered one
    as in the reference.  Returns (x, y, w, h).
    """
    H_img, W_img = frame.shape[:2]
    joints_2d, _ = estimator(frame, timestamp=timestamp)
    estimator.reset()
    x, y, w, h = bbox_update(joints_2d, W_img, H_img)
    if w < 1 or h < 1:
        return [0, 0, W_img, H_img]
    return [x, y, w, h]


def track(estimator, frames, rect=None, transpose=False, timestamps=None):
    """run_estimator_ps.py:80-109 without capture / drawing.  Yields (joints_2d, joints_3d, rect_used).

    frames: iterable of uint8 BGR arrays of one size; rect: initial (x, y, w, h), default the full frame;
    transpose: the reference's `T` option (np.rot90(frame, 3)); timestamps: optional iterable for reproducible
    filtering (default wall clock, like the reference).
    """
    ts = iter(timestamps) if timestamps is not None else None
    for frame in frames:
        if transpose:
            frame = np.rot90(frame, 3)
        H_img, W_img = frame.shape[:2]
        if rect is None:
            rect = [0, 0, W_img, H_img]
        x, y, w, h = rect
        if w < 1 or h < 1:  # a degenerate box (all joints on one pixel): fall back to the whole frame
            x, y, w, h = rect = [0, 0, W_img, H_img]
        frame_cropped = frame[y:y + h, x:x + w, :]
        joints_2d, joints_3d = estimator(frame_cropped, timestamp=next(ts) if ts is not None else None)
        joints_2d[:, 0] += y
        joints_2d[:, 1] += x
        used = [x, y, w, h]
        rect = bbox_update(joints_2d, W_img, H_img)
        yield joints_2d, joints_3d, used


def synthetic_stream(stream, n_frames, height=368, width=368, smooth=True):
    """Deterministic synthetic video: frame k of stream s has seed 1234 + 1000*s + k."""
    from .parallel import stream_seed
    from .weights import uniform01
    for k in range(n_frames):
        seed = stream_seed(stream, k)
        if not smooth:
            yield (uniform01(seed, height * width * 3) * 256).astype(np.uint8).reshape(height, width, 3)
            continue
        g = uniform01(seed, 9 * 9 * 3).reshape(9, 9, 3).astype(np.float64)
        ys, xs = np.linspace(0, 8, height, endpoint=False), np.linspace(0, 8, width, endpoint=False)
        y0, x0 = ys.astype(int), xs.astype(int)
        fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
        a = g[y0][:, x0] * (1 - fx) + g[y0][:, x0 + 1] * fx
        b = g[y0 + 1][:, x0] * (1 - fx) + g[y0 + 1][:, x0 + 1] * fx
        yield np.clip((a * (1 - fy) + b * fy) * 256, 0, 255).astype(np.uint8)


def main(argv=None):
    """python -m vnect_amd.runner [picture.jpg]: the run_pic.py flow on the MI355X, printing the 21 joints."""
    import os
    import sys
    argv = sys.argv[1:] if argv is None else argv
    from .estimator import VNectEstimator
    path = argv[0] if argv else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                              "tests", "golden", "test_pic.jpg")
    img = load_bgr(path)
    est = VNectEstimator()
    j2, j3, rect = run_pic(est, img)
    print("rect", rect)
    for i in range(21):
        print(i, j2[i], j3[i])
    est.close()


if __name__ == "__main__":
    main()
