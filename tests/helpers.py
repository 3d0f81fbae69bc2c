"""Shared deterministic input builders for the tests (no numpy RNG: counter-based hash only)."""
import numpy as np

from vnect_amd.weights import uniform01


def synth_frame(seed, h=368, w=368, smooth=False):
    """uint8 BGR frame.  smooth=True: 8x8 random grid bilinearly upsampled (structure for the resizes)."""
    if not smooth:
        return (uniform01(seed, h * w * 3) * 256).astype(np.uint8).reshape(h, w, 3)
    g = uniform01(seed, 9 * 9 * 3).reshape(9, 9, 3).astype(np.float64)
    ys = np.linspace(0, 8, h, endpoint=False)
    xs = np.linspace(0, 8, w, endpoint=False)
    y0, x0 = ys.astype(int), xs.astype(int)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    a = g[y0][:, x0] * (1 - fx) + g[y0][:, x0 + 1] * fx
    b = g[y0 + 1][:, x0] * (1 - fx) + g[y0 + 1][:, x0 + 1] * fx
    return np.clip((a * (1 - fy) + b * fy) * 256, 0, 255).astype(np.uint8)


def synth_maps(seed, S, amp=1.0):
    """(S,46,46,84) f32 network-output stand-in: hash noise + one Gaussian bump per joint and scale.

    Heatmap channels [0,21) get a bump of height ~1 at a seed-dependent cell (same cell for all scales,
    as a real pyramid would); location-map channels [21,84) are smooth ramps plus noise.
    """
    m = (uniform01(seed, S * 46 * 46 * 84).reshape(S, 46, 46, 84).astype(np.float64) - 0.5) * 0.1
    centers = (uniform01(seed + 7919, 42).reshape(21, 2) * 40 + 3)
    yy, xx = np.mgrid[0:46, 0:46].astype(np.float64)
    for j in range(21):
        cy, cx = centers[j]
        bump = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 1.5 ** 2))
        for s in range(S):
            m[s, :, :, j] += amp * bump * (1.0 + 0.05 * s)
            m[s, :, :, 21 + j] += (xx - 23) / 23.0 * (1 + 0.1 * j)
            m[s, :, :, 42 + j] += (yy - 23) / 23.0 * (1 - 0.02 * j)
            m[s, :, :, 63 + j] += np.sin(0.1 * (xx + yy) + j)
    return m.astype(np.float32)
