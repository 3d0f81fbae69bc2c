#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE'S OWN PYTHON.

Runs only in the build container (needs /root/reference); the .npz files it writes are what
travels.  Nothing here copies reference source: the reference modules are imported and driven.

  F1 oneeuro.npz      src/OneEuroFilter.py class, both estimator configs, irregular timestamps
  F2 readoff.npz      src/utils.py hm_pt_interp_bilinear / extract_3d_joints incl. edge regimes
  F3 glue_*.npz       src/estimator.py VNectEstimator.__call__ glue (lines 97-142) with TensorFlow and
                      cv2 replaced by stubs: sess.run returns scripted maps (tests/helpers.synth_maps),
                      cv2.resize is the oracle's restatement of OpenCV bilinear
  F4 net_samples.npz  oracle network on seeded weights/frame: sampled activations per stage,
                      cross-checked here against the independent torch-f64 restatement
  F6 angles.npz       src/joints2angles.py Joints2Angles (static formula + filtered __call__ with a scripted clock)
  F7 caffe_shapes.json  the OUTPUT CELLS of materials/caffe_script.ipynb (the authors' pycaffe session): blob shape of
                      every layer (cell 3) and shape of every parameter blob (cell 5) -- recorded data, the structural
                      known answers of the network
  test_pic.jpg        the reference's own data file pic/test_pic.jpg (data, 368 wide x 538 tall)
"""
import contextlib
import io
import os
import shutil
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

import oracle  # noqa: E402
from tests import helpers  # noqa: E402
from vnect_amd.weights import synthetic_weights, uniform01  # noqa: E402


# --------------------------------------------------------------------------- stubs
class FakeSession:
    """Stands in for tf.Session: run() returns the scripted maps of the current frame."""
    script = []     # list of (S,46,46,84) arrays, consumed in order
    fed = []        # batches the glue fed

    def run(self, fetches, feed_dict):
        (batch,) = feed_dict.values()
        FakeSession.fed.append(np.array(batch))
        m = FakeSession.script.pop(0)
        assert batch.shape[0] == m.shape[0]
        return [np.ascontiguousarray(m[..., 21 * q:21 * q + 21]) for q in range(4)]


def install_stubs():
    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR = 1

    def resize(img, dsize, fx=0, fy=0, interpolation=1):
        assert tuple(dsize) == (0, 0) and fx == fy and interpolation == 1
        return oracle.resize(np.ascontiguousarray(img), fx)

    cv2.resize = resize
    sys.modules["cv2"] = cv2
    tf = types.ModuleType("tensorflow")
    tf.Session = FakeSession
    saver = types.SimpleNamespace(restore=lambda sess, ckpt: None)
    tf.train = types.SimpleNamespace(import_meta_graph=lambda p: saver, latest_checkpoint=lambda p: "ckpt")
    tf.get_default_graph = lambda: types.SimpleNamespace(get_tensor_by_name=lambda n: n)
    sys.modules["tensorflow"] = tf
    sys.path.insert(0, os.path.join(REF, "src"))


# --------------------------------------------------------------------------- F1
def gen_oneeuro():
    from OneEuroFilter import OneEuroFilter
    out = {}
    cfgs = {"2d": dict(freq=30, mincutoff=1.7, beta=0.3, dcutoff=0.4),
            "3d": dict(freq=30, mincutoff=0.8, beta=0.4, dcutoff=0.4)}
    n = 64
    u = uniform01(101, 4 * n).astype(np.float64)
    # irregular but strictly increasing timestamps around 30 Hz, starting at a wall-clock-like value
    ts = 1.7e9 + np.cumsum(1 / 30 * (0.4 + 1.2 * u[:n]))
    xs = 184 + 60 * np.sin(np.arange(n) * 0.21) + 8 * (u[n:2 * n] - 0.5)
    for name, cfg in cfgs.items():
        f = OneEuroFilter(**cfg)
        out["y_" + name] = np.array([f(float(x), float(t)) for x, t in zip(xs, ts)])
    # timestamp 0.0 is falsy in `if self.__lasttime and timestamp` -> freq stays at its previous value
    ts0 = np.arange(n) / 30.0
    f = OneEuroFilter(**cfgs["2d"])
    out["y_t0"] = np.array([f(float(x), float(t)) for x, t in zip(xs, ts0)])
    out["ts"], out["ts0"], out["xs"] = ts, ts0, xs
    np.savez(os.path.join(HERE, "oneeuro.npz"), **out)
    assert out["y_2d"][0] == xs[0]  # first call is the identity


# --------------------------------------------------------------------------- F2
def gen_readoff():
    import utils
    maps = helpers.synth_maps(55, 1)[0].astype(np.float64)  # (46,46,84)
    xm, ym, zm = maps[..., 21:42], maps[..., 42:63], maps[..., 63:84]
    u = uniform01(202, 42).reshape(21, 2).astype(np.float64) * 367
    pts = u.copy()
    # edge regimes of hm_pt_interp_bilinear: src<0 extrapolation (p<3.5), src>=45 -> 0 (p>=363.5), exact cells
    pts[0] = [0.0, 0.0]
    pts[1] = [3.4, 367.0]
    pts[2] = [363.5, 2.0]
    pts[3] = [367.0, 367.0]
    pts[4] = [183.5, 183.5]
    pts[5] = [3.5, 363.49]
    pts[14] = [200.25, 150.75]
    single = np.array([utils.hm_pt_interp_bilinear(xm[:, :, j], 8, (pts[j, 0], pts[j, 1])) for j in range(21)])
    j3 = utils.extract_3d_joints(pts.copy(), xm, ym, zm, 8)
    assert j3.dtype == np.float32 and np.all(j3[14] == 0)
    np.savez(os.path.join(HERE, "readoff.npz"), pts=pts, single=single, joints_3d=j3, map_seed=55)


# --------------------------------------------------------------------------- F3
def gen_glue():
    import estimator as ref_est
    from PIL import Image

    shutil.copyfile(os.path.join(REF, "pic", "test_pic.jpg"), os.path.join(HERE, "test_pic.jpg"))
    pic = np.asarray(Image.open(os.path.join(HERE, "test_pic.jpg")).convert("RGB"))[:, :, ::-1].copy()  # BGR
    assert pic.shape == (538, 368, 3)
    cases = {
        "pic_default": dict(frames=[pic] * 3, scales=None, seed=900),              # h > w, reference scales
        "wide_default": dict(frames=[helpers.synth_frame(31 + k, 300, 500, smooth=True) for k in range(3)],
                             scales=None, seed=910),                               # h < w
        "square_baseline": dict(frames=[helpers.synth_frame(1234 + k) for k in range(4)],
                                scales=[1.0, 0.8, 0.6], seed=920),                 # BASELINE.json scales
        "square_one_scale": dict(frames=[helpers.synth_frame(77, smooth=True)], scales=[1], seed=930),
    }
    for name, c in cases.items():
        with contextlib.redirect_stdout(io.StringIO()):
            est = ref_est.VNectEstimator()
        if c["scales"] is not None:
            est.scales = c["scales"]
        S = len(est.scales)
        nfr = len(c["frames"])
        # scripted clock: __call__ reads time.time() at t0, in joint_filter(2d), joint_filter(3d), and for the FPS print
        t2d = 1.7e9 + np.arange(nfr) / 30.0 + 0.004
        t3d = t2d + 0.0007
        clock = []
        for k in range(nfr):
            # time.time() returns a Python float (an np.float64 here would change numpy's scalar promotion)
            clock += [float(t2d[k] - 0.004), float(t2d[k]), float(t3d[k]), float(t3d[k] + 0.001)]
        ref_est.time = types.SimpleNamespace(time=lambda: clock.pop(0))
        FakeSession.script = [helpers.synth_maps(c["seed"] + k, S) for k in range(nfr)]
        FakeSession.fed = []
        j2s, j3s, meta = [], [], []
        for k in range(nfr):
            with contextlib.redirect_stdout(io.StringIO()):
                j2, j3 = est(c["frames"][k])
                _, scaler, (ox, oy) = est.gen_input_batch(c["frames"][k], est.box_size, est.scales)
            assert j2.shape == (21, 2) and j2.dtype == np.float64 and j3.shape == (21, 3) and j3.dtype == np.float32
            j2s.append(j2.copy()), j3s.append(j3.copy()), meta.append([scaler, ox, oy])
        fed = np.stack(FakeSession.fed[:nfr])
        assert fed.shape == (nfr, S, 368, 368, 3) and fed.dtype == np.float32
        np.savez(os.path.join(HERE, "glue_%s.npz" % name),
                 joints_2d=np.stack(j2s), joints_3d=np.stack(j3s), meta=np.array(meta, np.float64),
                 t2d=t2d, t3d=t3d, scales=np.array(est.scales, np.float64), map_seed=c["seed"],
                 batch_sum=fed.astype(np.float64).sum(axis=(2, 3, 4)),
                 batch_probe=fed[:, :, ::37, ::41, :].copy())


# --------------------------------------------------------------------------- F4
def gen_net_samples():
    import torch
    from tests import torch_net
    w = synthetic_weights()
    frame = helpers.synth_frame(1234)
    batch, _, _ = oracle.gen_input_batch(frame, [1.0])
    net = oracle.Oracle(w, keep=True)
    out = net.forward(batch)
    taps = {}
    with torch.no_grad():
        ref = torch_net.forward(w, batch, taps=taps).numpy()
    assert np.abs(out - ref).max() <= 2e-5 * np.abs(ref).max()
    names = ["conv1", "pool1", "res2a", "res2c", "res3a", "res3d", "res4a", "res4f", "res5a", "res5b_branch2c_new",
             "res5c_branch1a", "res5c_branch2a", "res5c_branch2a_feat", "res5c_branch2b", "res5c_branch2c"]
    data = {}
    for i, n in enumerate(names):
        a = net.activation(n).ravel()
        idx = (uniform01(4000 + i, 256).astype(np.float64) * a.size).astype(np.int64)
        t = taps[n].numpy().ravel()
        assert np.abs(a[idx] - t[idx]).max() <= 2e-5 * np.abs(t).max(), n
        data["idx_" + n], data["val_" + n], data["f64_" + n] = idx, a[idx], t[idx]
    data["out_sum"] = np.float64(out.astype(np.float64).sum())
    data["out_abs_sum"] = np.float64(np.abs(out.astype(np.float64)).sum())
    np.savez(os.path.join(HERE, "net_samples.npz"), **data)


# --------------------------------------------------------------------------- F6
def gen_angles():
    import joints2angles as ref
    n = 24
    u = uniform01(606, n * 63 + n).astype(np.float64)
    joints = ((u[:n * 63] - 0.5) * 1200.0).reshape(n, 21, 3).astype(np.float32)  # mm, like extract_3d_joints' output
    ts = 1.7e9 + np.cumsum(1 / 30 * (0.5 + u[n * 63:]))
    static = np.array([ref.Joints2Angles.joints2angles(j) for j in joints])
    with contextlib.redirect_stdout(io.StringIO()):
        obj = ref.Joints2Angles(filter=True)
        clock = iter(np.repeat(ts, 8))          # __call__ reads time.time() once per angle (joints2angles.py:50)
        ref.time.time = lambda: float(next(clock))
        filtered = np.array([obj(j) for j in joints])
    np.savez(os.path.join(HERE, "angles.npz"), joints=joints, ts=ts, static=static, filtered=filtered)


# --------------------------------------------------------------------------- F7
def gen_caffe_shapes():
    import json
    import re
    nb = json.load(open(os.path.join(REF, "materials", "caffe_script.ipynb")))
    outs = ["".join("".join(o.get("text", [])) for o in c.get("outputs", [])) for c in nb["cells"]]
    blobs = {m.group(1): [int(x) for x in m.group(2).split(",")]
             for m in re.finditer(r"The output shape of layer (\S+) is \(([\d, ]+)\)", outs[3])}
    params, name = {}, None
    for line in outs[5].splitlines():
        if line.strip() and not line.startswith("\t"):
            name = line.strip()
            params[name] = []
        elif line.strip():
            params[name].append([int(x) for x in re.findall(r"\d+", line)])
    assert len(blobs) > 100 and len(params) == 56 and sum(len(v) for v in params.values()) == 110
    json.dump({"blobs": blobs, "params": params}, open(os.path.join(HERE, "caffe_shapes.json"), "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    install_stubs()
    gen_caffe_shapes()
    gen_angles()
    gen_oneeuro()
    gen_readoff()
    gen_glue()
    gen_net_samples()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
