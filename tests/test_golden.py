"""CPU: pin the oracle against fixtures recorded from the reference's own Python (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

import oracle
from tests import helpers

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    with np.load(os.path.join(G, name)) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("cfg,key,tkey", [
    (dict(freq=30, mincutoff=1.7, beta=0.3, dcutoff=0.4), "y_2d", "ts"),
    (dict(freq=30, mincutoff=0.8, beta=0.4, dcutoff=0.4), "y_3d", "ts"),
    (dict(freq=30, mincutoff=1.7, beta=0.3, dcutoff=0.4), "y_t0", "ts0"),
])
def test_oneeuro_bit_exact(cfg, key, tkey):
    """F1: src/OneEuroFilter.py sequences; float64 in, float64 out -> bit-exact."""
    g = _load("oneeuro.npz")
    f = oracle.OneEuro(**cfg)
    y = np.array([f(x, t) for x, t in zip(g["xs"], g[tkey])])
    assert np.array_equal(y, g[key])
    assert y[0] == g["xs"][0]


def test_oneeuro_equal_timestamps_raise():
    f = oracle.OneEuro(30, 1.7, 0.3, 0.4)
    f(1.0, 5.0)
    with pytest.raises(ZeroDivisionError):  # OneEuroFilter.py:66
        f(2.0, 5.0)


def test_readoff_bit_exact():
    """F2: utils.hm_pt_interp_bilinear / extract_3d_joints incl. the three edge regimes."""
    g = _load("readoff.npz")
    maps = helpers.synth_maps(int(g["map_seed"]), 1)[0].astype(np.float64)
    xm, ym, zm = maps[..., 21:42], maps[..., 42:63], maps[..., 63:84]
    single = np.array([oracle.hm_pt_interp(np.ascontiguousarray(xm[:, :, j]), 8, g["pts"][j]) for j in range(21)])
    assert np.array_equal(single, g["single"])
    j3 = oracle.extract_3d(g["pts"], xm, ym, zm)
    assert j3.dtype == np.float32 and np.array_equal(j3, g["joints_3d"])
    # src >= 45 (p >= 363.5): both taps are cell 45 and the weights cancel -> 0
    assert g["single"][3] == 0.0


@pytest.mark.parametrize("case", ["pic_default", "wide_default", "square_baseline", "square_one_scale"])
def test_call_glue_bit_exact(case):
    """F3: estimator.py:97-142 driven with stubbed TF/cv2; oracle must reproduce every frame bit for bit.

    The fixture was recorded under numpy 2.x (NEP 50 scalar promotion in the float32-fed 3-D filters),
    hence nep50=True; the numpy-1.x flavour is compared to it with a tolerance below.
    """
    g = _load("glue_%s.npz" % case)
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
    est = oracle.OracleEstimator(scales=scales, nep50=True)
    legacy = oracle.OracleEstimator(scales=scales, nep50=False)
    for k, frame in enumerate(frames):
        batch, scaler, (ox, oy) = oracle.gen_input_batch(frame, scales)
        assert [scaler, ox, oy] == list(g["meta"][k])
        assert np.array_equal(batch.astype(np.float64).sum(axis=(1, 2, 3)), g["batch_sum"][k])
        assert np.array_equal(batch[:, ::37, ::41, :], g["batch_probe"][k])
        maps = helpers.synth_maps(int(g["map_seed"]) + k, len(scales))
        j2, j3 = est.postprocess(maps, g["t2d"][k], g["t3d"][k], scaler, ox, oy)
        assert np.array_equal(j2, g["joints_2d"][k]), k
        assert np.array_equal(j3, g["joints_3d"][k]), k
        l2, l3 = legacy.postprocess(maps, g["t2d"][k], g["t3d"][k], scaler, ox, oy)
        assert np.array_equal(l2, j2)
        assert np.allclose(l3, j3, rtol=2e-5, atol=2e-3)  # f32- vs f64-state filters, << 0.05 mm


def test_net_samples_pinned(weights):
    """F4: oracle network output at sampled positions (regression pin; cross-checked vs torch f64 when made)."""
    g = _load("net_samples.npz")
    batch, _, _ = oracle.gen_input_batch(helpers.synth_frame(1234), [1.0])
    net = oracle.Oracle(weights, keep=True)
    out = net.forward(batch)
    for k in g:
        if k.startswith("idx_"):
            n = k[4:]
            a = net.activation(n).ravel()[g[k]]
            assert np.array_equal(a, g["val_" + n]), n
            assert np.abs(a - g["f64_" + n]).max() <= 2e-5 * max(np.abs(net.activation(n)).max(), 1e-6), n
    assert out.astype(np.float64).sum() == g["out_sum"]


def test_joints2angles_matches_reference_recording():
    """vnect_amd.angles.Joints2Angles vs values recorded from src/joints2angles.py (angles.npz):
    the static formula and the OneEuro-filtered __call__ with the recorded clock, bit for bit."""
    from vnect_amd.angles import Joints2Angles
    g = np.load(os.path.join(G, "angles.npz"))
    static = np.array([Joints2Angles.joints2angles(j) for j in g["joints"]])
    assert static.dtype == g["static"].dtype and np.array_equal(static, g["static"])
    obj = Joints2Angles(filter=True)
    filtered = np.array([obj(j, timestamp=float(t)) for j, t in zip(g["joints"], g["ts"])])
    assert np.array_equal(filtered, g["filtered"])
    assert np.array_equal(filtered[0], static[0])  # first call of a OneEuro filter is the identity
    plain = Joints2Angles(filter=False)
    assert np.array_equal(np.array(plain(g["joints"][3])), static[3])
