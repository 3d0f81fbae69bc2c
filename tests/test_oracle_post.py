"""CPU: oracle pre/post-processing against a second, numpy restatement and known answers."""
import numpy as np
import pytest

import oracle
from tests import helpers
from vnect_amd.weights import uniform01


def np_resize(img, f):
    """Independent numpy restatement of cv2.resize(img,(0,0),fx=f,fy=f,INTER_LINEAR) (OpenCV resize.cpp)."""
    img = np.asarray(img)
    sh, sw = img.shape[:2]
    dh, dw = int(np.rint(sh * f)), int(np.rint(sw * f))  # cvRound = round half to even
    if (dh, dw) == (sh, sw):
        return img.copy()
    scale = 1.0 / f
    x = img.reshape(sh, sw, -1)

    def table(ssize, dsize):
        fx = ((np.arange(dsize) + 0.5) * scale - 0.5).astype(np.float32)
        s0 = np.floor(fx).astype(np.int64)
        return s0, (fx - s0.astype(np.float32)).astype(np.float32)

    sx, fx = table(sw, dw)
    lo, hi = sx < 0, sx >= sw - 1
    fx = np.where(lo | hi, np.float32(0), fx)
    sx = np.clip(sx, 0, sw - 1)
    sx1 = np.minimum(sx + 1, sw - 1)
    sy, fy = table(sh, dh)
    sy0, sy1 = np.clip(sy, 0, sh - 1), np.clip(sy + 1, 0, sh - 1)
    if img.dtype == np.uint8:
        a1 = np.rint(fx * np.float32(2048)).astype(np.int64)
        a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
        b1 = np.rint(fy * np.float32(2048)).astype(np.int64)
        b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64)
        xi = x.astype(np.int64)
        rows = xi[:, sx] * a0[None, :, None] + xi[:, sx1] * a1[None, :, None]
        r0, r1 = rows[sy0], rows[sy1]
        out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
        out = out.astype(np.uint8)
    else:
        T = img.dtype.type
        a1, a0 = fx.astype(T), (np.float32(1) - fx).astype(T)
        rows = x[:, sx] * a0[None, :, None] + x[:, sx1] * a1[None, :, None]
        rows[:, hi] = x[:, sx][:, hi]
        b1, b0 = fy.astype(T), (np.float32(1) - fy).astype(T)
        out = rows[sy0] * b0[:, None, None] + rows[sy1] * b1[:, None, None]
    return out.reshape((dh, dw) + img.shape[2:])


@pytest.mark.parametrize("dtype", [np.uint8, np.float32, np.float64])
@pytest.mark.parametrize("shape,f", [((368, 368, 3), 0.8), ((368, 368, 3), 0.6), ((538, 368, 3), 368 / 538),
                                     ((46, 46, 21), 1 / 0.8), ((46, 46, 21), 1 / 0.7), ((46, 46), 8.0),
                                     ((37, 53, 3), 0.85), ((5, 7, 2), 3.3)])
def test_resize_two_restatements_agree(dtype, shape, f):
    n = int(np.prod(shape))
    u = uniform01(hash((shape, f)) & 0xFFFF, n).reshape(shape)
    img = (u * 256).astype(np.uint8) if dtype == np.uint8 else (u.astype(np.float64) * 4 - 2).astype(dtype)
    a, b = oracle.resize(img, f), np_resize(img, f)
    assert a.shape == b.shape and a.dtype == b.dtype
    assert np.array_equal(a, b)


def test_resize_known_answers():
    assert oracle.cvround(57.5) == 58 and oracle.cvround(56.5) == 56  # half to even: 46*1.25 -> 58
    assert oracle.resize(np.zeros((46, 46, 21), np.float32), 1.25).shape == (58, 58, 21)
    assert oracle.resize(np.zeros((46, 46, 21), np.float32), 1 / 0.6).shape == (77, 77, 21)
    assert oracle.resize(np.zeros((46, 46, 21), np.float32), 1 / 0.85).shape == (54, 54, 21)
    assert oracle.resize(np.zeros((46, 46, 21), np.float32), 1 / 0.7).shape == (66, 66, 21)
    for s, d in [(0.8, 294), (0.6, 221), (0.85, 313), (0.7, 258)]:
        assert oracle.resize(np.zeros((368, 368, 3), np.uint8), s).shape[0] == d
    c = np.full((20, 30, 3), 137, np.uint8)
    assert np.all(oracle.resize(c, 0.7) == 137) and np.all(oracle.resize(c, 1.9) == 137)
    img = helpers.synth_frame(5, 40, 40)
    assert np.array_equal(oracle.resize(img, 1.0), img)  # same size -> copy
    # x8 upsample of a ramp stays a ramp in the interior: src = (d+0.5)/8-0.5
    ramp = np.tile(np.arange(46, dtype=np.float64), (46, 1))
    up = oracle.resize(ramp, 8.0)
    assert up.shape == (368, 368)
    assert np.allclose(up[100, 4:364], (np.arange(4, 364) + 0.5) / 8 - 0.5)
    assert np.all(up[:, :4] == 0) and np.all(up[:, 364:] == 45)


def test_gen_input_batch_layout():
    frame = helpers.synth_frame(9, 538, 368, smooth=True)  # like pic/test_pic.jpg: 368 wide, 538 tall
    batch, scaler, (ox, oy) = oracle.gen_input_batch(frame, [1, 0.85, 0.7])
    assert batch.shape == (3, 368, 368, 3) and batch.dtype == np.float32
    assert scaler == 368 / 538 and oy == 0 and ox == 184 - oracle.cvround(368 * scaler) // 2
    assert batch.min() >= np.float32(-0.4) and batch.max() <= np.float32(1 - 0.4)
    w2 = oracle.cvround(368 * scaler)
    assert np.all(batch[0, :, :ox] == np.float32(-0.4)) and np.all(batch[0, :, ox + w2:] == np.float32(-0.4))
    # scale 0.7 -> 258 px image centred with 55 px of padding each side (368-258 = 110)
    assert np.all(batch[2, :55] == np.float32(-0.4)) and np.all(batch[2, 55 + 258:] == np.float32(-0.4))
    # odd remainder: 0.85 -> 313, pad 27 before and 28 after (utils.py:137-149)
    assert np.all(batch[1, :27] == np.float32(-0.4)) and np.all(batch[1, 27 + 313:] == np.float32(-0.4))
    assert not np.all(batch[1, 27] == np.float32(-0.4))


def test_merge_is_identity_for_single_unit_scale():
    maps = helpers.synth_maps(3, 1)
    avg = oracle.merge_scales(maps, [1.0])
    for q in range(4):
        assert np.array_equal(avg[q], maps[0, :, :, 21 * q:21 * q + 21].astype(np.float64))


def test_merge_matches_numpy_restatement():
    scales = [1.0, 0.8, 0.6]
    maps = helpers.synth_maps(4, 3)
    avg = oracle.merge_scales(maps, scales)
    ref = np.zeros((4, 46, 46, 21))
    for i, s in enumerate(scales):
        for q in range(4):
            r = np_resize(np.ascontiguousarray(maps[i, :, :, 21 * q:21 * q + 21]), 1.0 / s)
            mid = r.shape[0] // 2
            ref[q] += r[mid - 23:mid + 23, mid - 23:mid + 23]
    ref /= 3
    assert np.array_equal(avg, ref)


def test_extract_2d_planted_peaks():
    """F5: planted Gaussian peaks -> argmax lands in the 8x8 block of the planted cell."""
    hm = np.zeros((46, 46, 21))
    yy, xx = np.mgrid[0:46, 0:46]
    cells = [(3 + 2 * j, 40 - j) for j in range(21)]
    for j, (cy, cx) in enumerate(cells):
        hm[:, :, j] = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 1.5 ** 2))
    j2 = oracle.extract_2d(hm)
    for j, (cy, cx) in enumerate(cells):
        assert abs(j2[j, 0] - (cy * 8 + 3.5)) <= 0.5 and abs(j2[j, 1] - (cx * 8 + 3.5)) <= 0.5
    # ties: first maximum in row-major order (np.argmax)
    flat = np.zeros((46, 46, 21))
    assert np.all(oracle.extract_2d(flat) == 0)
    # brute-force check against numpy on a noisy map
    noisy = helpers.synth_maps(8, 1)[0, :, :, :21].astype(np.float64)
    j2 = oracle.extract_2d(noisy)
    for j in range(21):
        up = np_resize(np.ascontiguousarray(noisy[:, :, j]), 8.0)
        assert tuple(j2[j]) == np.unravel_index(np.argmax(up), up.shape)


def test_float_resize_agrees_with_torch_bilinear():
    """An implementation nobody here wrote: torch's CPU bilinear kernel (align_corners=False, no antialias) uses the same
    half-pixel formula as cv2.resize(INTER_LINEAR) -- src = (d + 0.5) * (1 / f) - 0.5, clamped at 0, second tap clamped at the
    last sample -- wherever the two agree on the OUTPUT SIZE (torch floors size * f, cv2 rounds it).  That covers the x8
    up-sampling of utils.extract_2d_joints (46 -> 368, float64: same taps and weights -- multiples of 1/16 -- and only the order
    of the two lerps differs, so the results agree to an ulp or two) and the 1 / 0.85 re-scaling of the reference's default pyramid (46 -> 54, float32).  cv2 itself cannot be run
    here (parity unpinned); this pins the restated formula against a third party at least."""
    import torch
    import torch.nn.functional as F
    rng = np.random.RandomState(11)
    m = rng.standard_normal((46, 46)).astype(np.float64)
    up = oracle.resize(m, 8.0)
    t = F.interpolate(torch.from_numpy(m)[None, None], scale_factor=8.0, mode="bilinear", align_corners=False,
                      recompute_scale_factor=False)[0, 0].numpy()
    assert up.shape == t.shape == (368, 368)
    assert float(np.abs(up - t).max()) <= 4 * np.finfo(np.float64).eps * float(np.abs(m).max())   # measured: 1 ulp
    assert np.array_equal(up[:, :4], np.repeat(up[:, :1], 4, axis=1)) and np.array_equal(t[:, :4], np.repeat(t[:, :1], 4, axis=1))  # left clamp: d < 4 -> sample 0
    m32 = rng.standard_normal((46, 46, 21)).astype(np.float32)
    f = 1.0 / 0.85
    a = oracle.resize(m32, f)
    t = F.interpolate(torch.from_numpy(m32).permute(2, 0, 1)[None], scale_factor=f, mode="bilinear", align_corners=False,
                      recompute_scale_factor=False)[0].permute(1, 2, 0).numpy()
    assert a.shape == t.shape == (54, 54, 21)
    # same taps and the same float32 weights up to the order of the two lerps (cv2: horizontal first; torch: one fused
    # expression): a few ulp of the blended values
    assert float(np.abs(a - t).max()) <= 1e-5 * float(np.abs(m32).max())   # measured: 6e-6
