"""Downstream consumer of the 3-D joints (SURVEY 8 f-row 4): 21x3 joints -> 8 arm angles for a Baxter robot.

Mirrors ``Joints2Angles`` of the reference (``src/joints2angles.py:23-109``): same class name, constructor
flag, ``__call__`` and static ``joints2angles``; angles are float64 radians in the order
``s0_l, s1_l, e0_l, e1_l, s0_r, s1_r, e0_r, e1_r``.  Host-side numpy (21 joints: nothing for a GPU to do).
Differences: no console print, and ``__call__`` takes an optional ``timestamp`` (the reference reads
``time.time()``, ``joints2angles.py:50``) so that runs are reproducible.
"""
import math
import time

import numpy as np


class _LowPass:
    """``LowPassFilter`` of src/OneEuroFilter.py:13-38."""

    def __init__(self):
        self.y = self.s = None

    def __call__(self, value, alpha):
        s = value if self.y is None else alpha * value + (1.0 - alpha) * self.s
        self.y, self.s = value, s
        return s


class OneEuro:
    """``OneEuroFilter`` of src/OneEuroFilter.py:41-75 (scalar, Python floats): first call is the identity, a
    falsy timestamp (None or 0.0) keeps the previous frequency, equal timestamps raise ZeroDivisionError."""

    def __init__(self, freq, mincutoff=1.0, beta=0.0, dcutoff=1.0):
        self.freq, self.mincutoff, self.beta, self.dcutoff = float(freq), float(mincutoff), float(beta), float(dcutoff)
        self.x, self.dx, self.lasttime = _LowPass(), _LowPass(), None

    def _alpha(self, cutoff):
        te = 1.0 / self.freq
        tau = 1.0 / (2 * math.pi * cutoff)
        return 1.0 / (1.0 + tau / te)

    def __call__(self, x, timestamp=None):
        if self.lasttime and timestamp:
            self.freq = 1.0 / (timestamp - self.lasttime)
        self.lasttime = timestamp
        prev = self.x.y
        dx = 0.0 if prev is None else (x - prev) * self.freq
        edx = self.dx(dx, self._alpha(self.dcutoff))
        cutoff = self.mincutoff + self.beta * math.fabs(edx)
        return self.x(x, self._alpha(cutoff))


def _angle(v1, v2):
    """cal_angle, src/joints2angles.py:6-8."""
    return np.arccos(np.dot(v1, v2) / (np.linalg.norm(v1) * np.linalg.norm(v2)))


class Joints2Angles:
    def __init__(self, filter=True):
        self.filter = filter
        if filter:  # src/joints2angles.py:35-41
            self.filter_angles = [OneEuro(freq=120, mincutoff=0.5, beta=0.5, dcutoff=1.0) for _ in range(8)]

    def __call__(self, joints_3d, timestamp=None):
        angles = list(self.joints2angles(joints_3d))
        if self.filter:
            t = time.time() if timestamp is None else timestamp
            angles = [f(a, t) for f, a in zip(self.filter_angles, angles)]
        return angles

    @staticmethod
    def joints2angles(joints_3d):
        """src/joints2angles.py:59-109: shoulder (s0, s1) and elbow (e0, e1) angles of both arms."""
        j = np.asarray(joints_3d)
        s2e_l, e2w_l = j[6] - j[5], j[7] - j[6]  # left shoulder -> elbow -> wrist
        s2e_r, e2w_r = j[3] - j[2], j[4] - j[3]
        across_l = j[2] - j[5]                   # left shoulder -> right shoulder
        across_r = -across_l
        down = [0, 1, 0]
        n_l, n_r = np.cross(s2e_l, down), np.cross(s2e_r, down)
        b_l, b_r = np.cross(s2e_l, e2w_l), np.cross(s2e_r, e2w_r)
        s0_l = np.pi * 3 / 4 - _angle(across_l, n_l)
        s1_l = np.pi / 2 - _angle(down, s2e_l)
        e0_l = -_angle(n_l, b_l)
        e1_l = _angle(s2e_l, e2w_l)
        s0_r = np.pi / 4 - _angle(across_r, n_r)
        s1_r = np.pi / 2 - _angle(down, s2e_r)
        e0_r = _angle(n_r, b_r)
        e1_r = _angle(s2e_r, e2w_r)
        # the reference's final offsets (":99-103")
        return s0_l - np.pi / 4, s1_l, e0_l, -e1_l, s0_r + np.pi / 4, -s1_r, e0_r, e1_r
