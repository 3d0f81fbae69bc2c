"""VNectEstimator: the reference's call surface (src/estimator.py:16-142) over libvnect_hip.so.

    from vnect_amd import VNectEstimator
    est = VNectEstimator()                       # reference defaults (scales [1, 0.85, 0.7])
    joints_2d, joints_3d = est(frame_bgr_uint8)  # (21,2) float64 [row, col], (21,3) float32 mm

Everything from gen_input_batch to the un-mapping runs on the MI355X in hand-written HIP kernels;
this file only marshals arguments.  Additive keyword arguments (not in the reference): ``scales``,
``weights``, ``seed``, ``device``, ``precision``, ``paper_res2c``, ``use_graph``, and
``timestamp=`` on ``__call__`` for reproducible filtering (default: wall clock, like the reference).
"""
import time

import numpy as np

from . import _native
from .weights import MASTER_SEED, check_schema, load_weights, synthetic_weights


class VNectEstimator:
    # class attributes of src/estimator.py:18-25
    box_size = 368
    hm_factor = 8
    joints_sum = 21
    joint_parents = [16, 15, 1, 2, 3, 1, 5, 6, 14, 8, 9, 14, 11, 12, 14, 14, 1, 4, 7, 10, 13]

    def __init__(self, scales=None, weights=None, seed=MASTER_SEED, device=0, precision="fp32", paper_res2c=False,
                 use_graph="auto", numpy_promotion="legacy", verbose=True, lanes=1):
        if verbose:
            print('Initializing VNect Estimator...')
        # src/estimator.py:32; "for faster loops, use less scales e.g. [1], [1, 0.7]"
        self._scales = [float(s) for s in (scales if scales is not None else [1, 0.85, 0.7])]
        if isinstance(weights, str):
            weights = load_weights(weights)
        elif weights is None:
            # no trained weights ship with the reference (models/*/README.md): seeded synthetic ones
            weights = synthetic_weights(seed)
        else:
            check_schema(weights)
        self._cfg = dict(device=device, precision={"fp32": _native.FP32, "bf16": _native.BF16, "fp32_split": _native.FP32_SPLIT}[precision],
                         paper_res2c=paper_res2c, use_graph=use_graph,
                         numpy_promotion={"legacy": 0, "nep50": 1}[numpy_promotion], lanes=lanes)
        self._submitted = 0
        self._weights = weights
        self._h = None
        self._open()
        self.verbose = verbose
        if verbose:
            print('VNect Estimator initialized.')

    def _open(self):
        if self._h is not None:
            self._h.close()
        self._h = _native.Handle(self._scales, **self._cfg)
        self._h.set_weights(self._weights)
        self._h.finalize()

    # `self.scales` is a plain attribute in the reference; assigning it re-plans the pyramid
    @property
    def scales(self):
        return list(self._scales)

    @scales.setter
    def scales(self, value):
        value = [float(s) for s in value]
        if len(value) == len(self._scales):
            self._h.set_scales(value)
            self._scales = value
        else:  # batch size of the conv stack changes: rebuild the handle (filters restart, as a new estimator would)
            self._scales = value
            self._open()

    @property
    def handle(self):
        return self._h

    @staticmethod
    def gen_input_batch(img_input, box_size, scales, _handle=None):
        """src/estimator.py:70-81 on the device.  Returns (batch (S,368,368,3) f32, scaler, [offset_x, offset_y])."""
        if box_size != 368:
            raise ValueError("box_size is fixed at 368 by the network")
        h = _handle
        own = h is None or h.num_scales != len(scales)
        if own:
            # a pre-processing-only handle: resize tables + the batch buffer, no weights, no launch plan
            h = _native.Handle(list(scales), preprocess_only=True)
        else:
            h.set_scales(scales)
        try:
            return h.preprocess(img_input)
        finally:
            if own:
                h.close()

    def joint_filter(self, joints, dim=2, timestamp=None):
        """src/estimator.py:83-95: the estimator's 2-D (``dim=2``) or 3-D OneEuro bank applied to ``joints`` IN PLACE; returns
        ``joints``.  The filter state lives on the device (the same banks ``__call__`` advances), so this runs one tiny
        kernel.  A float32 array is filtered the way the reference filters its float32 ``joints_3d`` (numpy scalar promotion,
        ``numpy_promotion=``), any other dtype in float64 like ``joints_2d``.  ``timestamp=`` is additive (default: one
        ``time.time()`` per call, like the reference)."""
        t = time.time() if timestamp is None else float(timestamp)
        if not isinstance(joints, np.ndarray):  # the reference indexes joints[i, 0]: anything but an array fails there too
            raise TypeError("joints must be a numpy array (it is filtered in place)")
        a = joints
        dim = 2 if dim == 2 else 3  # the reference's `else` branch takes every other value as 3 (dim=1, dim=5, ... included)
        if a.ndim != 2 or a.shape[0] < self.joints_sum or a.shape[1] < dim:  # where the reference's joints[i, 2] raises
            raise IndexError("joints must hold at least (%d, %d) values" % (self.joints_sum, dim))
        try:
            out = self._h.joint_filter(dim, a[:self.joints_sum, :dim], a.dtype == np.float32, t)
        except _native.VnectError as e:
            self._raise_like_reference(e)
        a[:self.joints_sum, :dim] = out  # in place; numpy casts to the array's dtype as `joints[i, 0] = ...` does
        return joints

    @staticmethod
    def _raise_like_reference(e):
        if e.code == _native.E_TIMESTAMP:
            raise ZeroDivisionError("float division by zero") from e  # src/OneEuroFilter.py:66
        if e.code == _native.E_TIMEORDER:  # negative freq -> alpha outside (0, 1]: LowPassFilter.__setAlpha, OneEuroFilter.py:19-23
            raise ValueError("alpha should be in (0.0, 1.0]: timestamp earlier than the previous one") from e
        raise e

    def postprocess(self, maps, timestamp=None, scaler=1.0, offset_x=0, offset_y=0):
        t2d, t3d = self._stamps(timestamp)
        try:
            return self._h.postprocess(maps, t2d, t3d, scaler, offset_x, offset_y)
        except _native.VnectError as e:
            self._raise_like_reference(e)

    def forward(self, batch):
        """sess.run equivalent (src/estimator.py:100-104): (S,368,368,3) -> (S,46,46,84)."""
        return self._h.forward(batch)

    @staticmethod
    def _stamps(timestamp):
        if timestamp is None:  # src/estimator.py:84 reads time.time() once per joint_filter call
            return time.time(), time.time()
        if isinstance(timestamp, (tuple, list)):
            return float(timestamp[0]), float(timestamp[1])
        return float(timestamp), float(timestamp)

    def __call__(self, img_input, timestamp=None):
        t0 = time.time()
        t2d, t3d = self._stamps(timestamp)
        try:
            joints_2d, joints_3d = self._h.infer(img_input, t2d, t3d)
        except _native.VnectError as e:
            self._raise_like_reference(e)
        if self.verbose:
            print('FPS: {:>2.2f}'.format(1 / max(time.time() - t0, 1e-9)))
        return joints_2d, joints_3d

    # -- pipelined use (additive; the reference's loop is strictly one frame at a time) --------------------------------
    def submit(self, img_input, timestamp=None):
        """Queue a frame and return at once; at most ``max(lanes, 2)`` may be in flight.  With ``lanes=2`` / ``3`` the frames
        overlap on the GPU (+29 % frames/s for one video stream at three lanes -- `pipelined_frames_per_s_per_gpu` against `value` in
        profiles/r06_bench_line.json -- at the price of their latency); ``collect()`` returns
        results in order and they are bit-identical to calling the estimator frame by frame."""
        t2d, t3d = self._stamps(timestamp)
        slot = self._submitted % 4
        try:
            self._h.upload_frame(slot, img_input)
            self._h.submit_resident(slot, t2d, t3d)
        except _native.VnectError as e:
            self._raise_like_reference(e)
        self._submitted += 1

    def collect(self):
        """(joints_2d, joints_3d) of the oldest frame in flight."""
        return self._h.collect()

    def frame_buffer(self, height, width, index=0):
        """Additive: a (height, width, 3) uint8 array in PINNED host memory (two exist, ``index`` 0 / 1).  Capture into it
        (``cap.read(buf)``, ``buf[...] = frame``) and pass it -- or a crop of it, as the tracking loop does -- to the estimator: the
        frame then crosses PCIe without the CPU copy a pageable array needs first."""
        return self._h.frame_buffer(index, height, width)

    def reset(self):
        self._h.reset_filters()

    def close(self):
        if self._h is not None:
            self._h.close()
            self._h = None
