"""CPU: the C oracle's network (oracle/vnect_net.c) against an independent torch float64 restatement."""
import numpy as np
import pytest
import torch

import oracle
from tests import torch_net
from vnect_amd import weights as W


def _frame_batch(seed, S=1):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 256, (S, 368, 368, 3)).astype(np.float32) / 255 - 0.4).astype(np.float32)


def test_schema_counts():
    sc = W.schema()
    assert len(sc) == 109  # 109 TF variables
    assert sum(int(np.prod(s)) for _, s in sc) == 14615936
    w = W.synthetic_weights()
    W.check_schema(w)
    w2 = W.synthetic_weights()
    assert all(np.array_equal(w[k], w2[k]) for k in w)  # bit-reproducible
    assert abs(float(w["conv1/weights"].max()) - np.sqrt(6 / 147)) < 1e-3


def test_sgemm_matches_numpy():
    rng = np.random.default_rng(1)
    for (M, N, K) in [(7, 5, 3), (100, 33, 300), (529, 84, 128), (97, 191, 1024)]:
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = rng.standard_normal((K, N)).astype(np.float32)
        Cm = np.empty((M, N), np.float32)
        f = oracle.oracle.c_f32p
        oracle.lib().vo_sgemm(M, N, K, A.ctypes.data_as(f), K, B.ctypes.data_as(f), N, Cm.ctypes.data_as(f), N)
        ref = A.astype(np.float64) @ B.astype(np.float64)
        bound = (np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)).max()
        assert np.abs(Cm - ref).max() <= 2e-6 * bound


@pytest.mark.parametrize("paper", [False, True])
def test_network_vs_torch_f64(weights, paper):
    batch = _frame_batch(7)
    net = oracle.Oracle(weights, keep=True, paper_res2c=paper)
    out = net.forward(batch)
    taps = {}
    with torch.no_grad():
        ref = torch_net.forward(weights, batch, paper_res2c=paper, taps=taps).numpy()
    assert out.shape == (1, 46, 46, 84)
    # fp32 oracle vs f64: SURVEY 8(c) tolerance 1e-4 * max|map|; measured ~1e-6
    assert np.abs(out - ref).max() <= 2e-5 * np.abs(ref).max()
    # every stage, not only the end
    for name, t in taps.items():
        a = net.activation(name)
        t = t.numpy()
        assert a.shape == t.shape, name
        assert np.abs(a - t).max() <= 2e-5 * max(np.abs(t).max(), 1e-6), name
    # layer shapes of the caffe blob dump (materials/caffe_script.ipynb cell 3): 368->184->92->46->23->46
    assert net.activation("conv1").shape == (1, 184, 184, 64)
    assert net.activation("pool1").shape == (1, 92, 92, 64)
    assert net.activation("res3a_branch1").shape == (1, 46, 46, 512)
    assert net.activation("res4a_branch1").shape == (1, 23, 23, 1024)
    assert net.activation("res5c_branch2a_feat").shape == (1, 46, 46, 212)


def test_res2c_quirk_matters(weights):
    batch = _frame_batch(3)
    a = oracle.Oracle(weights).forward(batch)
    b = oracle.Oracle(weights, paper_res2c=True).forward(batch)
    assert np.abs(a - b).max() > 1e-3  # the wiring changes the result; default follows vnect_model.py:56


def test_batch_rows_independent(weights):
    """S images are independent through the net (what pyramid sharding relies on)."""
    batch = _frame_batch(11, S=2)
    net = oracle.Oracle(weights)
    both = net.forward(batch)
    one = net.forward(batch[1:2])
    assert np.array_equal(both[1:2], one)


def test_shapes_match_the_authors_caffe_session(weights):
    """Known answers recorded in the reference: materials/caffe_script.ipynb holds the OUTPUT of the authors' pycaffe
    session -- the blob shape of every layer (cell 3) and the shape of every parameter blob (cell 5); both are committed
    as data in tests/golden/caffe_shapes.json.  Every activation the oracle names must have the caffe blob's shape, and
    the weight schema must be the caffe parameters after caffe2pkl's transposes (src/caffe2pkl.py:48,60-80)."""
    import json
    import os
    from vnect_amd.weights import schema
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "caffe_shapes.json")))
    blobs, params = g["blobs"], g["params"]
    net = oracle.Oracle(weights, keep=True)
    net.forward(_frame_batch(5))
    checked = 0
    for blob, (c, hh, ww) in blobs.items():
        for name in (blob, blob + "_new"):  # caffe LAYERS res5a/5b carry a _new suffix that their blobs do not
            try:
                a = net.activation(name)
            except Exception:
                continue
            assert a.shape == (1, hh, ww, c), (name, a.shape, (c, hh, ww))
            checked += 1
            break
    assert checked >= 50, checked  # every activation the oracle names (54: convs, block outputs, deconvs, the 212-channel feature)
    assert blobs["x_heatmap"] == [21, 46, 46] and net.forward(_frame_batch(5)).shape == (1, 46, 46, 4 * 21)
    sch = dict(schema())
    n_arrays = 0
    for layer, shapes in params.items():
        if layer == "bn5c_branch2a":        # caffe BatchNorm: mean, variance, scale factor (folded by caffe2pkl.py:74-75)
            assert shapes == [[128], [128], [1]]
            assert sch[layer + "/moving_mean"] == (128,) and sch[layer + "/moving_variance"] == (128,)
            n_arrays += 2
        elif layer == "scale5c_branch2a":   # caffe Scale: gamma, beta
            assert shapes == [[128], [128]]
            assert sch["bn5c_branch2a/gamma"] == (128,) and sch["bn5c_branch2a/beta"] == (128,)
            n_arrays += 2
        elif len(shapes) == 1:              # bias-free conv / deconv -> '<scope>/kernel'
            o, i, kh, kw = shapes[0]
            want = (kh, kw, i, o) if layer == "res5c_branch2c" else (kh, kw, i, o)
            # caffe stores a Deconvolution as (Cin, Cout, kh, kw): the same (2,3,1,0) transpose yields (kh, kw, Cout, Cin)
            assert sch[layer + "/kernel"] == want, (layer, sch[layer + "/kernel"], want)
            n_arrays += 1
        else:                               # conv with bias: (Cout, Cin, kh, kw) -> (kh, kw, Cin, Cout)
            o, i, kh, kw = shapes[0]
            assert sch[layer + "/weights"] == (kh, kw, i, o) and sch[layer + "/biases"] == tuple(shapes[1]), layer
            n_arrays += 2
    assert n_arrays == len(sch) == 109
