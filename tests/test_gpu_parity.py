"""GPU (MI355X): parity of the HIP path, called through the C ABI, against the CPU oracle.

Tolerances: conv stack fp32 vs oracle fp32: max|d| <= 1e-4 * max|map| (measured ~1e-6);
pre-processing and post-processing on identical inputs: bit-exact; joints_2d equal or within the tie
rule; joints_3d |d| <= 0.05 mm + 1e-4*|v|.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
BASELINE_SCALES = [1.0, 0.8, 0.6]
T0 = 1.7e9


def _native():
    from vnect_amd import _native
    return _native


def _handle(scales, weights, **kw):
    n = _native()
    h = n.Handle(scales, **kw)
    h.set_weights(weights)
    h.finalize()
    return h


@pytest.fixture(scope="module")
def h3(weights):
    h = _handle(BASELINE_SCALES, weights)
    yield h
    h.close()


@pytest.fixture(scope="module")
def ref3(weights, oracle_net):
    import oracle
    return oracle.OracleEstimator(scales=BASELINE_SCALES, net=oracle_net)


def _log(name, obj):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, indent=1)


# ------------------------------------------------------------------------------------------ conv stack
def test_conv_stack_every_layer(h3, weights, oracle_net):
    """a1-a7: every named activation of the HIP graph vs the oracle, S=3 at BASELINE scales.  Inner layers are read from
    a handle with private buffers (keep_activations); the product default shares an arena between layers, must refuse to
    return an inner layer, and must produce bit-identical maps."""
    import oracle
    from tests import helpers
    batch, _, _ = oracle.gen_input_batch(helpers.synth_frame(1234, smooth=True), BASELINE_SCALES)
    ref = oracle_net.forward(batch)
    arena_out = h3.forward(batch)
    with pytest.raises(_native().VnectError):
        h3.activation("res3a")
    assert np.array_equal(h3.activation("res5c_branch2c"), arena_out)  # the final maps stay readable
    h3 = _handle(BASELINE_SCALES, weights, keep_activations=True)
    out = h3.forward(batch)
    assert np.array_equal(out, arena_out)
    rows, bad = [], []
    names = [L["name"] for L in h3.layers()]
    acts = ["conv1", "pool1"]
    for n in names:
        if "+" in n:                          # two layers in one launch: "<scope_a>+<rest of scope_b>"
            a, b = n.split("+")
            acts += [a, b if b.startswith("res") else a.split("_")[0] + "_" + b]
        elif n == "res5c_branch2c":
            acts.append(n)
        elif n.endswith("_branch2c") or n == "res5a_branch2c_new":
            acts.append(n.split("_")[0])      # block output resNx
        elif n == "res5c_deconv":
            continue
        elif n == "res5c_bone_length":
            acts.append("res5c_branch2a_feat")
        elif n not in ("conv1", "pool1"):
            acts.append(n)
    for n in acts:
        a, r = h3.activation(n), oracle_net.activation(n)
        assert a.shape == r.shape, n
        err = float(np.abs(a - r).max() / max(np.abs(r).max(), 1e-12))
        rows.append((n, list(a.shape), err))
        if not err <= 1e-4:
            bad.append((n, err))
    _log("layer_errors.json", rows)
    for r in rows:
        print("%-24s %-20s rel err %.3g" % (r[0], r[1], r[2]))
    h3.close()
    assert not bad, "first mismatching layers: %r" % bad[:3]
    assert float(np.abs(out - ref).max() / np.abs(ref).max()) <= 1e-4


def test_conv1_span_form_is_bit_identical(weights, oracle_net, monkeypatch):
    """conv1 (fp32) reads its A operand from the tile's contiguous pixel run(s) instead of 64 gathered windows and skips the MFMAs
    of the zero padding channel (conv.hip, SPAN): same K order, so conv1 and everything behind it must equal the gathered-window
    form (VNECT_NO_SPAN=1) bit for bit -- on frames whose tiles straddle output rows and images (S = 1, 2, 3)."""
    import oracle
    from tests import helpers
    for scales in ([1.0], [1.0, 0.7], BASELINE_SCALES):
        batch, _, _ = oracle.gen_input_batch(helpers.synth_frame(640 + len(scales), 333, 368), scales)
        h = _handle(scales, weights, keep_activations=True, use_graph=False)
        out = h.forward(batch)
        c1 = h.activation("conv1")
        monkeypatch.setenv("VNECT_NO_SPAN", "1")
        ref_out = h.forward(batch)
        ref_c1 = h.activation("conv1")
        monkeypatch.delenv("VNECT_NO_SPAN")
        h.close()
        assert np.array_equal(c1, ref_c1), len(scales)
        assert np.array_equal(out, ref_out), len(scales)
        r = oracle_net.forward(batch)
        assert float(np.abs(out - r).max()) <= 1e-4 * float(np.abs(r).max())


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_tail_and_chain_gemms_are_bit_identical(weights, monkeypatch, prec):
    """The launches that carry a second (and third) GEMM behind their K loop -- res2*_branch2b>branch2c on 64x64 tiles, res3*_branch2b>
    branch2c and the head's res5c_branch2b>res5c_branch2c on 32x128 tiles (conv.hip: tail_gemm, tail_wide), in bf16 with the next
    block's branch2a chained on (chain_gemm) --, and the stem launch that also runs res2a_branch2a + res2a_branch1 on its pooled tile
    (stem.hip, PAIR) -- against the same layers as launches of their own (vnect_model.py:32-103,211-217): the final maps must be EQUAL
    at 3 scales (199 workgroups) and at 4 (the wide form no longer fits one workgroup per CU and the plan falls back by itself) and
    agree to rounding at 1 and 2 (67 / 133 workgroups; there the stand-alone layers split K), and the launch counts must be what the
    plan promises.  VNECT_FORCE_CHAIN puts the chain on the fp32 handle too (measured slower there, so off by default -- but it is
    built, so it is tested)."""
    import oracle
    from tests import helpers
    n = _native()
    p = n.BF16 if prec == "bf16" else n.FP32
    for scales in ([1.0], [1.0, 0.7], BASELINE_SCALES, [1, 0.85, 0.7, 0.5]):
        batch, _, _ = oracle.gen_input_batch(helpers.synth_frame(77 + len(scales), smooth=True), scales)
        outs, counts, joints = {}, {}, {}
        frame = helpers.synth_frame(91 + len(scales), 368, 300, smooth=True)
        # (fp32 plans take the wide form from 128 workgroups on -- it is slower for a single scale's 67 --: VNECT_FORCE_WIDE_TAIL keeps
        # that geometry under test)
        monkeypatch.setenv("VNECT_FORCE_WIDE_TAIL", "1")
        for tag, env in (("default", {}), ("no_wide", {"VNECT_NO_WIDE_TAIL": "1"}), ("no_tail", {"VNECT_NO_TAIL": "1"}),
                         ("no_chain", {"VNECT_NO_CHAIN": "1"}), ("force_chain", {"VNECT_FORCE_CHAIN": "1"}),
                         ("no_stem_pair", {"VNECT_NO_STEM_PAIR": "1"})):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            h = _handle(scales, weights, precision=p)
            for k in env:
                monkeypatch.delenv(k)
            outs[tag] = h.forward(batch)
            if tag in ("default", "no_stem_pair"):  # the stem's pair GEMM from the frame as well (forward() feeds it the batch tensor)
                j2, j3 = h.infer(frame, T0 + 5, T0 + 5.001)
                joints[tag] = (j2, j3, h.activation("res5c_branch2c"))
            counts[tag] = sum(1 for L in h.layers() if ">" in L["name"]), sum(L["name"].count(">") for L in h.layers())
            h.close()
        S = len(scales)
        for x, y in zip(joints["default"], joints["no_stem_pair"]):
            assert np.array_equal(x, y), scales
        for tag in outs:
            if S >= 3:  # the stand-alone layers run whole-K tiles like the fused ones: the same sums in the same order
                assert np.array_equal(outs[tag], outs["no_tail"]), (scales, tag)
            else:       # fewer scales: stand-alone 3x3 (and, for one scale, 1x1) layers split K (hostplan.h: choose_tile), so only the order of the sums differs
                err = float(np.abs(outs[tag] - outs["no_tail"]).max() / np.abs(outs["no_tail"]).max())
                assert err <= (2e-2 if prec == "bf16" else 1e-5), (scales, tag, err)
        assert counts["no_tail"] == (0, 0)
        assert counts["no_wide"] == ((3, 4 if prec == "bf16" else 3) if S <= 3 else (0, 0))  # the 92x92 tails: up to 512 tiles of 64 rows
        wide = S <= 3
        assert counts["no_chain"] == ((8, 8) if wide else (0, 0))
        assert counts["force_chain"] == ((8, 12 if prec == "bf16" else 11) if wide else (0, 0))  # bf16: the 64-wide tail of res2a chains too
        assert counts["default"] == counts["no_stem_pair"] == counts["force_chain" if prec == "bf16" else "no_chain"]
        monkeypatch.delenv("VNECT_FORCE_WIDE_TAIL")
        if S == 1 and prec == "fp32":  # the plan's own choice for one scale in fp32: the 92x92 tails only
            h = _handle(scales, weights, precision=p)
            assert sum(1 for L in h.layers() if ">" in L["name"]) == 3
            assert np.array_equal(h.forward(batch), outs["no_wide"])
            h.close()


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_fused_stem_is_bit_identical(weights, monkeypatch, prec):
    """The stem as ONE launch (stem.hip: [gen_input_batch ->] conv1 + ReLU -> max-pool on spatial tiles, vnect_model.py:27-29,
    estimator.py:70-81) against the three stand-alone kernels: pool1 -- read from handles with private buffers, VNECT_FORCE_STEM puts
    the fused form on such a handle -- the final maps and the joints must be EQUAL, for both patch sources (the batch tensor; the
    uint8 frame), S = 1, 3 and 4 (row groups of 4 and of 5 pooled rows; more tiles than CUs), square and non-square frames, and
    through vnect_forward's (S,368,368,3) entry."""
    import oracle
    from tests import helpers
    n = _native()
    p = n.BF16 if prec == "bf16" else n.FP32
    # long side 368 (squarify is a copy: the from-the-frame form runs, with and without black bars), and frames that squarify resizes
    # (that form's host-side fallback: pyramid_kernel + the stem from the batch tensor)
    frames = [helpers.synth_frame(31, smooth=True), helpers.synth_frame(32, 538, 368, smooth=True), helpers.synth_frame(33, 240, 320),
              helpers.synth_frame(34, 368, 200), helpers.synth_frame(35, 123, 368, smooth=True), helpers.synth_frame(36)]
    # [1.0, 0.8, 0.3], [1.0, 0.15]: scales whose tiles need more frame rows than the kernel's LDS rectangle holds (plan::stem_frame_fits
    # -> fallback); one and two images run 2- and 3-row tiles (hostplan.h: stem_groups), three 4- and 5-row ones
    for scales in (BASELINE_SCALES, [1.0], [1, 0.85, 0.7, 0.5], [1.0, 0.4], [1.0, 0.3], [1.0, 0.8, 0.3], [1.0, 0.15]):
        plain = _handle(scales, weights, precision=p, keep_activations=True)
        assert [L["name"] for L in plain.layers()][:2] == ["conv1", "pool1"]
        batch, _, _ = oracle.gen_input_batch(frames[1], scales)
        want_fwd = plain.forward(batch)
        want_pool_fwd = plain.activation("pool1")
        want = []
        for k, f in enumerate(frames):
            t = T0 + 900 + k / 30
            j2, j3 = plain.infer(f, t, t + 0.001)
            want.append((j2, j3, plain.activation("pool1"), plain.activation("res5c_branch2c")))
        plain.close()
        for mode in ("batch", "frame"):
            monkeypatch.setenv("VNECT_FORCE_STEM", mode)
            fused = _handle(scales, weights, precision=p, keep_activations=True)
            monkeypatch.delenv("VNECT_FORCE_STEM")
            assert np.array_equal(fused.forward(batch), want_fwd), (scales, mode)          # vnect_forward: the stem reads the batch
            assert np.array_equal(fused.activation("pool1"), want_pool_fwd), (scales, mode)
            for k, f in enumerate(frames):
                t = T0 + 900 + k / 30
                j2, j3 = fused.infer(f, t, t + 0.001)
                assert np.array_equal(fused.activation("pool1"), want[k][2]), (scales, mode, k)
                assert np.array_equal(fused.activation("res5c_branch2c"), want[k][3]), (scales, mode, k)
                assert np.array_equal(j2, want[k][0]) and np.array_equal(j3, want[k][1]), (scales, mode, k)
            fused.close()
    # the product default: arena handles run the stem from the frame; VNECT_NO_STEM restores the three launches
    a = _handle(BASELINE_SCALES, weights, precision=p)
    monkeypatch.setenv("VNECT_NO_STEM", "1")
    b = _handle(BASELINE_SCALES, weights, precision=p)
    monkeypatch.delenv("VNECT_NO_STEM")
    for k, f in enumerate(frames):
        t = T0 + 950 + k / 30
        ra, rb = a.infer(f, t, t + 0.001), b.infer(f, t, t + 0.001)
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]), k
        assert np.array_equal(a.activation("res5c_branch2c"), b.activation("res5c_branch2c")), k
    a.close(), b.close()


def test_fused_stem_fuzz_frame_shapes_and_scales(weights, monkeypatch):
    """The from-the-frame stem against the three stand-alone kernels on a sweep of frame shapes whose long side is 368 (its fast path:
    odd widths and heights, so frame rows start at every byte alignment; black bars left / right or top / bottom) and of scale sets
    (1 to 4 scales down to 0.5, incl. a scale whose size rounds back to 368): the final maps and the joints must be EQUAL."""
    from tests import helpers
    rng = np.random.RandomState(2024)
    shapes = [(368, 368), (368, 1), (1, 368), (368, 367), (367, 368), (368, 123), (77, 368), (368, 245), (201, 368)]
    scale_sets = [[1.0], [1.0, 0.8, 0.6], [1, 0.85, 0.7], [1.0, 0.9999], [1.0, 0.93, 0.71, 0.5], [0.9, 0.55]]
    for si, scales in enumerate(scale_sets):
        monkeypatch.setenv("VNECT_NO_STEM", "1")
        ref = _handle(scales, weights)
        monkeypatch.delenv("VNECT_NO_STEM")
        fused = _handle(scales, weights)
        for k in rng.choice(len(shapes), 4, replace=False):
            H, W = shapes[k]
            frame = helpers.synth_frame(5000 + 17 * si + int(k), H, W, smooth=bool((si + k) % 2))
            t = T0 + 2000 + si * 10 + int(k)
            fused.reset_filters(), ref.reset_filters()
            a2, a3 = fused.infer(frame, t, t + 0.001)
            b2, b3 = ref.infer(frame, t, t + 0.001)
            assert np.array_equal(fused.activation("res5c_branch2c"), ref.activation("res5c_branch2c")), (scales, H, W)
            assert np.array_equal(a2, b2) and np.array_equal(a3, b3), (scales, H, W)
        fused.close(), ref.close()


def test_deconv_three_accumulator_shape(weights, oracle_net, monkeypatch):
    """The transposed convs res5c_branch1a / res5c_branch2a (+ BN + ReLU; vnect_model.py:188-196) on 64 x 96 tiles with two K groups and THREE
    accumulators per wave (conv.hip: NACC; round 4: 200 tiles in one round instead of 300 in two) -- the fp32 plan at three scales.  Plain
    (per-layer read-back) and with the bone features in its launch (arena plan): res5c_branch2a_feat within 1e-4 of the oracle, the two forms
    bit-identical, the 64 x 64 plan (VNECT_NO_DECONV96=1) equal to fp32 rounding (another summation order: two K groups), final maps alike;
    bf16, four scales and a split-product handle keep the 64 x 64 tiles."""
    import oracle
    from tests import helpers
    n = _native()
    batch, _, _ = oracle.gen_input_batch(helpers.synth_frame(1234, smooth=True), BASELINE_SCALES)
    ref = oracle_net.forward(batch)
    feat_ref = oracle_net.activation("res5c_branch2a_feat")
    top = float(np.abs(feat_ref).max())

    def tile_of(h):
        L = [x for x in h.layers() if x["name"].startswith("res5c_deconv")]
        assert len(L) == 1
        return (L[0]["tile_m"], L[0]["tile_n"], L[0]["workgroups"], L[0]["name"])

    keep = _handle(BASELINE_SCALES, weights, keep_activations=True)
    arena = _handle(BASELINE_SCALES, weights)
    assert tile_of(keep)[:3] == (64, 96, 200) and tile_of(arena) == (64, 96, 200, "res5c_deconv+bone_length"), (tile_of(keep), tile_of(arena))
    mk, ma = keep.forward(batch), arena.forward(batch)
    fk = keep.activation("res5c_branch2a_feat")
    assert fk.shape == feat_ref.shape and float(np.abs(fk - feat_ref).max()) <= 1e-4 * top
    assert np.array_equal(mk, ma)                                  # bone features inside the launch == the stand-alone bone kernel
    assert float(np.abs(ma - ref).max()) <= 1e-4 * float(np.abs(ref).max())
    monkeypatch.setenv("VNECT_NO_DECONV96", "1")
    old = _handle(BASELINE_SCALES, weights, keep_activations=True)
    monkeypatch.delenv("VNECT_NO_DECONV96")
    assert tile_of(old)[:3] == (64, 64, 300)
    mo = old.forward(batch)
    fo = old.activation("res5c_branch2a_feat")
    assert float(np.abs(fo - fk).max()) <= 2e-5 * top and float(np.abs(mo - mk).max()) <= 2e-5 * float(np.abs(ref).max())
    for h in (keep, arena, old):
        h.close()
    for kw, scales in ((dict(precision=n.BF16), BASELINE_SCALES), (dict(precision=n.FP32_SPLIT), BASELINE_SCALES), (dict(), [1.0, 0.9, 0.8, 0.7])):
        h = _handle(scales, weights, **kw)
        assert tile_of(h)[:2] == (64, 64), (kw, scales, tile_of(h))
        h.close()


def test_conv_stack_batch_independent(h3, oracle_net):
    """The S images are independent: permuting the batch permutes the output (what sharding relies on)."""
    import oracle
    from tests import helpers
    batch, _, _ = oracle.gen_input_batch(helpers.synth_frame(99), BASELINE_SCALES)
    a = h3.forward(batch)
    b = h3.forward(batch[::-1].copy())
    assert np.array_equal(a, b[::-1])
    assert np.array_equal(a, h3.forward(batch))  # deterministic: bit-identical on a second run


def test_single_scale_and_paper_wiring(weights):
    """S=1 (reference's 'faster loops' hint) and the paper_res2c switch, vs the oracle."""
    import oracle
    from tests import helpers
    batch, _, _ = oracle.gen_input_batch(helpers.synth_frame(5, smooth=True), [1.0])
    for paper in (False, True):
        h = _handle([1.0], weights, paper_res2c=paper)
        ref = oracle.Oracle(weights, paper_res2c=paper).forward(batch)
        out = h.forward(batch)
        h.close()
        assert float(np.abs(out - ref).max() / np.abs(ref).max()) <= 1e-4, paper


@pytest.mark.parametrize("force", ["64,64,1,1", "64,32,2,1", "32,32,4,1", "64,64,1,5", "64,32,2,2", "32,32,4,3"])
@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp32_split"])
def test_every_tile_shape_on_every_layer(weights, oracle_net, monkeypatch, force, prec):
    """The launch plan picks a tile shape per layer (64x64, 64x32 x 2 K groups, 32x32 x 4 K groups, 5-way split-K); here
    every shape is FORCED onto every layer that admits it (VNECT_FORCE_TILE = BM,BN,KG,ks), so each kernel variant --
    including K groups combined with cross-workgroup slabs -- sees 1x1, 3x3, strided, transposed and 7x7 layers, S = 2."""
    import oracle
    from tests import helpers
    scales = [1.0, 0.7]
    batch, _, _ = oracle.gen_input_batch(helpers.synth_frame(77, smooth=True), scales)
    ref = oracle_net.forward(batch)
    monkeypatch.setenv("VNECT_FORCE_TILE", force)
    monkeypatch.setenv("VNECT_NO_STEM", "1")   # conv1 as a launch of its own (the fused stem has ONE shape; its parity test is below)
    h = _handle(scales, weights, precision={"bf16": _native().BF16, "fp32_split": _native().FP32_SPLIT}.get(prec, _native().FP32))
    shapes = {(L["tile_m"], L["tile_n"], L["split_k"]) for L in h.layers() if L["M"]}
    out = h.forward(batch)
    again = h.forward(batch)
    h.close()
    bm, bn, kg, ks = (int(x) for x in force.split(","))
    assert (bm, bn) in {(a, b) for a, b, _ in shapes}, shapes  # the forced shape is really in the plan
    err = float(np.abs(out - ref).max() / np.abs(ref).max())
    print(force, prec, "rel err %.3g" % err, sorted(shapes))
    assert err <= (3e-2 if prec == "bf16" else 1e-4)   # the split-product path is held to the fp32 gate
    assert np.array_equal(out, again)  # K-group and slab sums run in a fixed order


# ------------------------------------------------------------------------------------------ pre-processing
@pytest.mark.parametrize("shape,smooth", [((368, 368), False), ((538, 368), True), ((300, 500), True),
                                          ((720, 1280), True), ((97, 61), False), ((368, 367), False)])
@pytest.mark.parametrize("scales", [BASELINE_SCALES, [1, 0.85, 0.7]])
def test_preprocess_bit_exact(weights, shape, smooth, scales):
    """a10: gen_input_batch on the device == oracle, bit for bit (8-bit fixed-point bilinear + pad + /255-0.4)."""
    import oracle
    from tests import helpers
    frame = helpers.synth_frame(hash(shape) & 0xFFF, shape[0], shape[1], smooth=smooth)
    h = _handle(scales, weights)
    batch, scaler, off = h.preprocess(frame)
    rb, rs, roff = oracle.gen_input_batch(frame, scales)
    h.close()
    assert scaler == rs and off == roff
    assert np.array_equal(batch, rb)


def test_preprocess_fuzz_shapes(h3):
    """a10 over 32 seeded frame shapes (long side 40 .. 1500, aspect ratios up to 6:1, odd sizes): every one bit-exact against the
    oracle, or rejected by both (utils.py:98-103 cannot place a crop whose scaled long side is not 368)."""
    import oracle
    from tests import helpers
    from vnect_amd._native import VnectError
    rng = np.random.RandomState(20240807)
    checked = 0
    for k in range(32):
        long_side = int(rng.randint(40, 1501))
        short = max(8, int(long_side / rng.uniform(1.0, 6.0)))
        H, W = (long_side, short) if k & 1 else (short, long_side)
        frame = helpers.synth_frame(500 + k, H, W, smooth=bool(k & 2))
        try:
            rb, rs, roff = oracle.gen_input_batch(frame, BASELINE_SCALES)
        except Exception:
            with pytest.raises(VnectError):
                h3.preprocess(frame)
            continue
        b, s, off = h3.preprocess(frame)
        assert s == rs and off == roff, (H, W)
        assert np.array_equal(b, rb), (H, W)
        checked += 1
    assert checked >= 24


def test_preprocess_strided_crop(h3):
    """Callers pass crops of a larger frame (run_estimator_ps.py:87): row stride != 3*W."""
    import oracle
    from tests import helpers
    big = helpers.synth_frame(3, 480, 640, smooth=True)
    crop = big[40:400, 100:420]
    b, s, off = h3.preprocess(crop)
    rb, rs, roff = oracle.gen_input_batch(np.ascontiguousarray(crop), BASELINE_SCALES)
    assert s == rs and off == roff and np.array_equal(b, rb)


# ------------------------------------------------------------------------------------------ post-processing
@pytest.mark.parametrize("promo", [0, 1])
def test_postprocess_bit_exact_sequence(weights, promo):
    """a12-a16 on identical maps over 6 frames (filters engaged, irregular dt): bit-exact vs the oracle."""
    import oracle
    from tests import helpers
    h = _handle(BASELINE_SCALES, weights, numpy_promotion=promo)
    ref = oracle.OracleEstimator(scales=BASELINE_SCALES, nep50=bool(promo))
    t = T0
    for k in range(6):
        maps = helpers.synth_maps(300 + k, 3)
        t += 1 / 30 + 0.003 * (k % 3)
        a2, a3 = h.postprocess(maps, t, t + 0.0007, 368 / 538, 58, 0)
        r2, r3 = ref.postprocess(maps, t, t + 0.0007, 368 / 538, 58, 0)
        assert np.array_equal(a2, r2), k
        assert np.array_equal(a3, r3), k
    h.close()


def test_four_scales_whole_path(weights, oracle_net):
    """More than three scales takes the general (8-scale) form of the merge code and a 4-image batch through the conv stack:
    pre-processing and post-processing bit-exact, final maps and joints within the fp32 tolerances."""
    import oracle
    from tests import helpers
    scales = [1.0, 0.9, 0.75, 0.6]
    h = _handle(scales, weights)
    ref = oracle.OracleEstimator(scales=scales, net=oracle_net)
    frame = helpers.synth_frame(321, 400, 310, smooth=True)
    b, sc, off = h.preprocess(frame)
    rb, rs, roff = oracle.gen_input_batch(frame, scales)
    assert np.array_equal(b, rb) and sc == rs and off == roff
    maps, rmaps = h.forward(b), oracle_net.forward(rb)
    assert np.abs(maps - rmaps).max() <= 1e-4 * np.abs(rmaps).max()
    for k in range(3):
        t = T0 + k / 30
        a2, a3 = h.postprocess(rmaps, t, t + 0.0005, sc, off[0], off[1])
        r2, r3 = ref.postprocess(rmaps, t, t + 0.0005, sc, off[0], off[1])
        assert np.array_equal(a2, r2) and np.array_equal(a3, r3), k
    h.close()


def test_six_scales_take_the_unfused_plans(weights, oracle_net):
    """The fused forms of the conv launch (tail GEMM at 92x92, bone features inside the transposed conv) need one tile per
    workgroup; with six scales the 92x92 layers have 794 tiles and the transposed conv 600 items, so the stand-alone layers, the
    dual-output 3x3 launch and the bone kernel run instead (and every tile streams).  Same tolerance against the oracle, and the
    five-scale handle (fused transposed conv, unfused 92x92) agrees too."""
    import oracle
    from tests import helpers
    frame = helpers.synth_frame(2024, 300, 368, smooth=True)
    for scales in ([1.0, 0.95, 0.9, 0.8, 0.7, 0.6], [1.0, 0.9, 0.8, 0.7, 0.6]):
        h = _handle(scales, weights)
        names = [L["name"] for L in h.layers()]
        assert "res2b_branch2b+res2c_branch2b" in names and not any(">" in n for n in names), names
        assert ("res5c_bone_length" in names) == (len(scales) == 6), names
        rb, _, _ = oracle.gen_input_batch(frame, scales)
        b, _, _ = h.preprocess(frame)
        assert np.array_equal(b, rb)
        out, ref = h.forward(b), oracle_net.forward(rb)
        assert float(np.abs(out - ref).max()) <= 1e-4 * float(np.abs(ref).max()), len(scales)
        h.close()


@pytest.mark.parametrize("promo", [0, 1])
def test_postprocess_long_filter_chain(weights, promo):
    """The OneEuro state is a recurrence: 150 frames of moving peaks (a drifting mixture of 5 map sets, so every joint's
    arg-max wanders and the read-off crosses cell borders), irregular frame times, and a timestamp 0.0 ("no timestamp",
    OneEuroFilter.py:65) in the middle -- still bit-exact against the oracle on every frame, no drift."""
    import oracle
    from tests import helpers
    scales = [1.0, 0.7]
    h = _handle(scales, weights, numpy_promotion=promo)
    ref = oracle.OracleEstimator(scales=scales, nep50=bool(promo))
    base = [helpers.synth_maps(700 + k, 2) for k in range(5)]
    t = T0
    for k in range(150):
        w = 0.5 + 0.5 * np.sin(0.13 * k + np.arange(5))
        maps = sum(float(wi) * b for wi, b in zip(w, base)).astype(np.float32)
        t += 1 / 30 + 0.004 * ((k * 7) % 5)
        t2d = 0.0 if k == 70 else t
        a2, a3 = h.postprocess(maps, t2d, t + 0.0004, 1.0, 0, 0)
        r2, r3 = ref.postprocess(maps, t2d, t + 0.0004, 1.0, 0, 0)
        assert np.array_equal(a2, r2) and np.array_equal(a3, r3), k
    h.close()


@pytest.mark.parametrize("case", ["pic_default", "wide_default", "square_baseline", "square_one_scale"])
def test_postprocess_reproduces_reference_recordings(weights, case):
    """The device pre+post-processing reproduces what the reference's own Python returned (fixtures F3)."""
    from tests import helpers
    with np.load(os.path.join(G, "glue_%s.npz" % case)) as z:
        g = {k: z[k] for k in z.files}
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
    h = _handle(scales, weights, numpy_promotion=1)  # fixtures were recorded under numpy 2.x
    for k, frame in enumerate(frames):
        batch, scaler, (ox, oy) = h.preprocess(frame)
        assert [scaler, ox, oy] == list(g["meta"][k])
        assert np.array_equal(batch.astype(np.float64).sum(axis=(1, 2, 3)), g["batch_sum"][k])
        assert np.array_equal(batch[:, ::37, ::41, :], g["batch_probe"][k])
        maps = helpers.synth_maps(int(g["map_seed"]) + k, len(scales))
        j2, j3 = h.postprocess(maps, g["t2d"][k], g["t3d"][k], scaler, ox, oy)
        assert np.array_equal(j2, g["joints_2d"][k]), k
        assert np.array_equal(j3, g["joints_3d"][k]), k
    h.close()


def test_argmax_ties_and_planted_peaks(weights):
    h = _handle([1.0], weights)
    flat = np.zeros((1, 46, 46, 84), np.float32)
    j2, j3 = h.postprocess(flat, T0, T0)
    assert np.all(j2 == 0) and np.all(j3 == 0)  # first maximum in row-major order
    hm = np.zeros((1, 46, 46, 84), np.float32)
    yy, xx = np.mgrid[0:46, 0:46]
    cells = [(3 + 2 * j, 40 - j) for j in range(21)]
    for j, (cy, cx) in enumerate(cells):
        hm[0, :, :, j] = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 1.5 ** 2))
    h.reset_filters()
    j2, _ = h.postprocess(hm, T0, T0)
    for j, (cy, cx) in enumerate(cells):
        assert abs(j2[j, 0] - (cy * 8 + 3.5)) <= 0.5 and abs(j2[j, 1] - (cx * 8 + 3.5)) <= 0.5
    h.close()


# ------------------------------------------------------------------------------------------ end to end
class _EndToEnd:
    """The end-to-end gate, applied to EVERY frame and EVERY joint (no allowance for a fraction of mismatching joints).

    A frame's result is split into the two things that can differ from the CPU oracle:
      (1) the conv stack's final maps: |GPU - oracle| <= 1e-4 * max|oracle| (fp32 summation order);
      (2) everything after them -- merge, arg-max, both filter banks, read-off, un-mapping -- which is exact arithmetic:
          the GPU's joints must equal, BIT FOR BIT, the oracle's post-processing of the GPU's own maps (a second oracle
          estimator whose filters advance in lockstep).
    Where (1) moves a heat-map maximum, np.argmax may legally pick another cell: the tie rule (utils.py:153-175 takes the first
    maximum) accepts a different raw arg-max only if the ORACLE's upsampled heat-map at the GPU's arg-max is within the
    map tolerance of its maximum.  A joint whose raw arg-max has agreed on every frame so far has comparable filter
    state, and is additionally held to joints_2d equal (1e-6) and joints_3d within 0.05 mm + 1e-4 * |v| of the full
    oracle chain."""

    def __init__(self, scales, oracle_net, nep50=False):
        import oracle
        self.scales, self.net = scales, oracle_net
        self.full = oracle.OracleEstimator(scales=scales, net=oracle_net, nep50=nep50)   # oracle maps -> oracle joints
        self.post = oracle.OracleEstimator(scales=scales, nep50=nep50)                   # GPU maps -> oracle joints
        self.clean = np.ones(21, bool)  # raw arg-max equal on every frame so far
        self.ties = 0
        self.worst3 = 0.0

    def check(self, frame, t2d, t3d, j2, j3, gpu_maps, tag=""):
        import oracle
        frame = np.ascontiguousarray(frame)
        batch, scaler, (ox, oy) = oracle.gen_input_batch(frame, self.scales)
        ref_maps = self.net.forward(batch)
        top = float(np.abs(ref_maps).max())
        assert float(np.abs(gpu_maps - ref_maps).max()) <= 1e-4 * top, tag                      # (1)
        p2, p3 = self.post.postprocess(gpu_maps, t2d, t3d, scaler, ox, oy)
        assert np.array_equal(j2, p2) and np.array_equal(j3, p3), tag                          # (2) bit for bit
        r2, r3 = self.full.postprocess(ref_maps, t2d, t3d, scaler, ox, oy)  # == the oracle's whole __call__ on this frame
        avg_ref = oracle.merge_scales(ref_maps, self.scales)[0]
        raw_ref = oracle.extract_2d(avg_ref)
        raw_gpu = oracle.extract_2d(oracle.merge_scales(gpu_maps, self.scales)[0])
        for j in range(21):                                                                    # tie rule, every joint
            if np.array_equal(raw_gpu[j], raw_ref[j]):
                continue
            up = oracle.resize(np.ascontiguousarray(avg_ref[:, :, j]), 8.0)
            assert up[int(raw_gpu[j, 0]), int(raw_gpu[j, 1])] >= up.max() - 1e-4 * top, (tag, j)
            self.clean[j] = False
            self.ties += 1
        c = self.clean
        assert np.all(np.abs(j2[c] - r2[c]) <= 1e-6 / min(scaler, 1.0) + 1e-9), tag
        d3 = np.abs(j3 - r3)
        # the root joint (14) is subtracted from every row: rows are comparable only while joint 14 is clean too
        if c[14]:
            tol = 0.05 + 1e-4 * np.abs(r3)
            assert np.all(d3[c] <= tol[c]), tag
            if c.any():
                self.worst3 = max(self.worst3, float((d3[c] - tol[c]).max()))


def test_end_to_end_vs_oracle(h3, oracle_net):
    """Whole __call__ over 4 frames against the oracle: see _EndToEnd (every frame, every joint)."""
    from tests import helpers
    h3.reset_filters()
    e2e = _EndToEnd(BASELINE_SCALES, oracle_net)
    for k in range(4):
        frame = helpers.synth_frame(1234 + k, smooth=True)
        t = T0 + k / 30
        j2, j3 = h3.infer(frame, t, t + 0.001)
        e2e.check(frame, t, t + 0.001, j2, j3, h3.activation("res5c_branch2c"), k)
    print("legal arg-max ties: %d, worst 3-D excess over tolerance: %.3g" % (e2e.ties, e2e.worst3))


def test_end_to_end_nonsquare_frames(h3, oracle_net):
    """Whole __call__ on frames that are not 368x368 (the size of pic/test_pic.jpg, a landscape VGA-like crop, a small portrait
    one): squarify scaler and centring offsets enter the un-mapping (estimator.py:137-139).  Same gate as the square case."""
    from tests import helpers
    h3.reset_filters()
    e2e = _EndToEnd(BASELINE_SCALES, oracle_net)
    for k, (H, W) in enumerate([(538, 368), (240, 320), (200, 120), (538, 368)]):
        frame = helpers.synth_frame(4321 + k, H, W, smooth=True)
        t = T0 + 100 + k / 30
        j2, j3 = h3.infer(frame, t, t + 0.001)
        e2e.check(frame, t, t + 0.001, j2, j3, h3.activation("res5c_branch2c"), (H, W))


@pytest.mark.parametrize("scales", [[1, 0.85, 0.7], [1.0, 0.7], [1.0]])
def test_end_to_end_reference_default_scales(weights, oracle_net, scales):
    """Whole __call__ at the reference's own pyramid (estimator.py:32: [1, 0.85, 0.7]) and at the two shorter ones its comment
    suggests "for faster loops" ([1, 0.7], [1]), on frames of the test picture's size: the every-frame, every-joint gate."""
    from tests import helpers
    h = _handle(scales, weights)
    e2e = _EndToEnd([float(s) for s in scales], oracle_net)
    for k in range(3):
        frame = helpers.synth_frame(3100 + k, 538, 368, smooth=True)
        t = T0 + 200 + k / 30 + 0.002 * k
        j2, j3 = h.infer(frame, t, t + 0.0013)
        e2e.check(frame, t, t + 0.0013, j2, j3, h.activation("res5c_branch2c"), (scales, k))
    h.close()


@pytest.mark.parametrize("lanes,graph", [(1, True), (2, True), (2, False), (3, True)])
def test_pipelined_submit_collect_equals_sequential(weights, lanes, graph):
    """Two frames in flight (submit k+1 before collecting k) return exactly what one-at-a-time inference returns: on one
    lane (same stream), and on two lanes (lanes=2: the frames overlap on two streams / activation arenas and only the
    joints kernels -- the OneEuro filter chain -- are ordered by an event).  Frames of different sizes alternate, so each
    lane keeps its own crop geometry; 12 frames exercise both lanes and the 4-deep result ring several times."""
    from tests import helpers
    shapes = [(368, 368), (300, 420), (368, 368), (410, 260)]
    frames = [helpers.synth_frame(500 + k, *shapes[k], smooth=True) for k in range(4)]
    a = _handle(BASELINE_SCALES, weights, lanes=lanes, use_graph=graph)
    b = _handle(BASELINE_SCALES, weights, use_graph=False)
    for k, f in enumerate(frames):
        a.upload_frame(k, f)
        b.upload_frame(k, f)
    n = 12
    seq = [b.infer_resident(k % 4, T0 + k / 30, T0 + k / 30 + 0.001) for k in range(n)]
    depth = max(lanes, 2)  # frames kept in flight
    got = []
    for k in range(n):
        if k >= depth:
            got.append(a.collect())
        a.submit_resident(k % 4, T0 + k / 30, T0 + k / 30 + 0.001)
    for _ in range(depth):
        got.append(a.collect())
    with pytest.raises(_native().VnectError):
        a.collect()  # nothing left in flight
    for k, ((g2, g3), (s2, s3)) in enumerate(zip(got, seq)):
        assert np.array_equal(g2, s2) and np.array_equal(g3, s3), k  # bit for bit
    # back to one at a time on the same handle: the filter chain continues across the mode change
    k = n
    g2, g3 = a.infer_resident(1, T0 + k / 30, T0 + k / 30 + 0.001)
    s2, s3 = b.infer_resident(1, T0 + k / 30, T0 + k / 30 + 0.001)
    assert np.array_equal(g2, s2) and np.array_equal(g3, s3)
    a.close(), b.close()


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp32_split"])
def test_soak_three_lanes_deterministic(weights, prec):
    """Race detector: 3 000 frames of one stream, three in flight on three lanes, twice, and once frame by frame -- all
    three result sequences must be identical bit for bit (the K-group hand-off through LDS, the lane events, the result
    ring and the arena sharing all have to be right every single time for that)."""
    from tests import helpers
    frames = [helpers.synth_frame(40 + k, smooth=(k % 2 == 0)) for k in range(8)]
    h = _handle(BASELINE_SCALES, weights, lanes=3, num_frame_slots=8,
                precision={"bf16": _native().BF16, "fp32_split": _native().FP32_SPLIT}.get(prec, _native().FP32))
    for k, f in enumerate(frames):
        h.upload_frame(k, f)
    n = 3000

    def run(depth):
        h.reset_filters()
        out = np.empty((n, 21, 5), np.float64)
        got = 0
        for k in range(n):
            if k >= depth:
                j2, j3 = h.collect()
                out[got, :, :2], out[got, :, 2:] = j2, j3
                got += 1
            h.submit_resident((k * 3) % 8, T0 + k / 30, T0 + k / 30 + 0.0005)
        while got < n:
            j2, j3 = h.collect()
            out[got, :, :2], out[got, :, 2:] = j2, j3
            got += 1
        return out

    a, b, c = run(3), run(3), run(1)
    h.close()
    assert np.all(np.isfinite(a))
    assert np.array_equal(a, b), "two pipelined runs differ at frame %d" % int(np.argmax(np.any(a != b, axis=(1, 2))))
    assert np.array_equal(a, c), "pipelined and frame-by-frame runs differ at frame %d" % int(np.argmax(np.any(a != c, axis=(1, 2))))


def test_estimator_submit_collect_two_lanes(weights):
    """The facade's additive pipelined API (submit / collect, lanes=2) against its own frame-by-frame __call__, with frames
    uploaded from host memory each time (an upload waits only for the inference that still reads its slot)."""
    from tests import helpers
    from vnect_amd import VNectEstimator
    frames = [helpers.synth_frame(900 + k, 368 - 11 * (k % 3), 300 + 17 * (k % 4), smooth=True) for k in range(9)]
    one = VNectEstimator(scales=BASELINE_SCALES, weights=weights, verbose=False)
    two = VNectEstimator(scales=BASELINE_SCALES, weights=weights, verbose=False, lanes=2)
    want = [one(f, timestamp=T0 + k / 30) for k, f in enumerate(frames)]
    got = []
    two.submit(frames[0], timestamp=T0)
    for k in range(1, len(frames)):
        two.submit(frames[k], timestamp=T0 + k / 30)
        got.append(two.collect())
    got.append(two.collect())
    for k, ((g2, g3), (w2, w3)) in enumerate(zip(got, want)):
        assert np.array_equal(g2, w2) and np.array_equal(g3, w3), k
    with pytest.raises(_native().VnectError):  # a third frame in flight is refused, state untouched
        two.submit(frames[0], timestamp=T0 + 1)
        two.submit(frames[1], timestamp=T0 + 2)
        two.submit(frames[2], timestamp=T0 + 3)
    one.close(), two.close()


def test_errors_mirror_reference(weights):
    from vnect_amd import VNectEstimator
    est = VNectEstimator(scales=[1.0], weights=weights, verbose=False)
    frame = np.zeros((368, 368, 3), np.uint8)
    est(frame, timestamp=5.0)
    with pytest.raises(ZeroDivisionError):   # OneEuroFilter.py:66
        est(frame, timestamp=5.0)
    with pytest.raises(ValueError):
        est(np.zeros((368, 368), np.uint8))
    j2, j3 = est(frame, timestamp=6.0)
    assert j2.shape == (21, 2) and j2.dtype == np.float64 and j3.shape == (21, 3) and j3.dtype == np.float32
    est.scales = [1.0, 0.7]   # assignable like the reference attribute
    j2, j3 = est(frame, timestamp=7.0)
    assert np.all(np.isfinite(j2))
    est.close()


def test_tracking_loop_variable_crops(weights, oracle_net):
    """run_estimator_ps.py:80-109 headless: the crop changes every frame, so squarify/resize tables are rebuilt per
    call; every frame and every joint is checked against the oracle fed the same crop (_EndToEnd)."""
    from vnect_amd import VNectEstimator, runner
    scales = [1.0, 0.8, 0.6]
    est = VNectEstimator(scales=scales, weights=weights, verbose=False)
    e2e = _EndToEnd(scales, oracle_net)
    frames = list(runner.synthetic_stream(3, 4, 480, 640))
    rect, sizes = [40, 30, 500, 400], set()
    for k, frame in enumerate(frames):
        x, y, w, h = rect
        crop = frame[y:y + h, x:x + w, :]
        sizes.add(crop.shape)
        t = T0 + k / 30
        j2, j3 = est(crop, timestamp=(t, t + 0.001))
        e2e.check(crop, t, t + 0.001, j2, j3, est.handle.activation("res5c_branch2c"), k)
        j2[:, 0] += y
        j2[:, 1] += x
        rect = runner.bbox_update(j2, 640, 480)
        if rect[2] < 8 or rect[3] < 8:
            rect = [0, 0, 640, 480]
    assert len(sizes) >= 2   # the loop really exercised more than one crop geometry
    est.close()


def test_tracking_loop_follows_planted_blobs():
    """The tracking loop with a KNOWN answer on the GPU (tests/planted.py; CPU twin with the oracle: tests/test_planted.py): a 640 x 480 video of
    three drifting blobs through runner.track (run_estimator_ps.py:80-109) -- whole frame first, then crops by the box rule, every crop a
    different size, squarified and resized on the device.  fp32: the loop is the ORACLE's loop joint for joint (joints_2d array_equal in every
    frame -- planted peaks leave no ties --, hence the same crops; joints_3d within the fp32 tolerance); fp32 and bf16: every joint within two
    heat-map cells of its blob, every crop within 2.5 cells + 8 pixels of the box rule applied to the true positions."""
    import oracle
    from tests import planted
    from tests.test_planted import moving_person
    from vnect_amd import VNectEstimator, runner
    H, W, n = 480, 640, 8
    pw = planted.weights()
    net = oracle.Oracle(pw)
    frames = [planted.scene(H, W, moving_person(k), sigma=10.0, seed=k) for k in range(n)]
    stamps = [T0 + 50 + i / 30 for i in range(n)]

    class OracleEst:
        def __init__(self):
            self.o = oracle.OracleEstimator(scales=BASELINE_SCALES, net=net)

        def __call__(self, img, timestamp=None):
            return self.o(np.ascontiguousarray(img), timestamp, timestamp)

    ref = list(runner.track(OracleEst(), frames, timestamps=stamps))
    for prec in ("fp32", "bf16"):
        est = VNectEstimator(scales=BASELINE_SCALES, weights=pw, precision=prec, verbose=False)
        prev_ideal, worst = None, 0.0
        for k, (j2, j3, rect) in enumerate(runner.track(est, frames, timestamps=stamps)):
            want = np.array([moving_person(k)[j % 3][:2] for j in range(21)], np.float64)
            cell = 8.0 / (368.0 / max(rect[2], rect[3]))
            worst = max(worst, float(np.abs(j2 - want).max()) / cell)
            assert np.abs(j2 - want).max() <= max(2.0 * cell, 14.0), (prec, k, rect, float(np.abs(j2 - want).max()))
            if prev_ideal is not None:
                assert np.abs(np.array(rect) - np.array(prev_ideal)).max() <= 2.5 * prev_cell + 8, (prec, k, rect, prev_ideal)
            prev_ideal, prev_cell = runner.bbox_update(want, W, H), cell
            if prec == "fp32":
                r2, r3, rrect = ref[k]
                assert rect == rrect and np.array_equal(j2, r2), (k, rect, rrect)
                assert np.all(np.abs(j3 - r3) <= 0.05 + 1e-4 * np.abs(r3)), k
        print("%s: tracked %d frames, joints at most %.2f heat-map cells from their blobs" % (prec, n, worst))
        est.close()


def test_planted_joints_through_every_way_of_running_a_frame():
    """The geometric known answer of tests/planted.py (joint j ON the bright blob of colour j % 3) through every way the library runs a frame
    -- each of them bit-equal to the synchronous call elsewhere in this file; here they must also be RIGHT: frames in flight on three lanes,
    three videos on one handle, the split-product path, a pyramid sharded over three rank handles (each builds and runs ONE scale; the maps
    are stacked in rank order as the exchange delivers them -- a swapped or stale slot would pull the merged peaks off the blobs), and one
    scale alone.  Tolerance: one box pixel (the oracle's own: tests/test_planted.py)."""
    from tests import planted
    n = _native()
    pw = planted.weights()
    frames = [planted.frame(700 + k, H, W) for k, (H, W) in enumerate([(368, 368), (538, 368), (240, 320), (368, 368)])]

    def on_blobs(j2, k, tol=1.0):
        frame, centres = frames[k]
        scaler = 368.0 / max(frame.shape[:2])
        d = float(np.abs(j2 - planted.expected(centres)).max())
        assert d <= tol / scaler, (k, d, scaler)

    # frames in flight on three lanes (a new stream per frame would reset the filters; one stream: the blobs jump between frames, so
    # only the first frame is unfiltered -- use one handle per check instead and submit the SAME frame three times: the filters settle on it)
    h = _handle(BASELINE_SCALES, pw, lanes=3)
    for k in range(3):
        h.upload_frame(k, frames[0][0])
        h.submit_resident(k, T0 + k / 30, T0 + k / 30 + 0.001)
    for k in range(3):
        on_blobs(h.collect()[0], 0)
    # three videos on one handle: stream s sees frame s
    h.reset_filters()
    for s_ in range(3):
        h.upload_frame(s_, frames[s_][0])
        h.submit_stream(s_, s_, T0 + 10 + s_, T0 + 10 + s_ + 0.001)
    for s_ in range(3):
        st, j2, j3 = h.collect_stream()
        on_blobs(j2, st)
    h.close()
    # the split-product path and a single scale
    for kw, scales in ((dict(precision=n.FP32_SPLIT), BASELINE_SCALES), (dict(), [1.0]), (dict(precision=n.BF16), [1.0, 0.8])):
        h = _handle(scales, pw, **kw)
        for k in range(len(frames)):
            h.reset_filters()
            on_blobs(h.infer(frames[k][0], T0 + 20 + k, T0 + 20 + k + 0.001)[0], k, tol=8.0 if kw.get("precision") == n.BF16 else 1.0)
        h.close()
    # pyramid sharded over three rank handles: rank r pre-processes and runs scale r; the stack in rank order is what the exchange delivers
    ranks = [n.Handle(BASELINE_SCALES, pyramid=(r, 3)) for r in range(3)]
    for hh in ranks:
        hh.set_weights(pw)
        hh.finalize()
    for k in range(len(frames)):
        maps = []
        for hh in ranks:
            b, scaler, (ox, oy) = hh.preprocess(frames[k][0])
            maps.append(hh.forward(b)[0])
        ranks[0].reset_filters()
        j2, j3 = ranks[0].postprocess(np.stack(maps), T0 + 30 + k, T0 + 30 + k + 0.001, scaler, ox, oy)
        on_blobs(j2, k)
        if k == 1:  # the wrong slot order is NOT right: the test can see what it claims to see
            ranks[0].reset_filters()
            w2, _ = ranks[0].postprocess(np.stack([maps[1], maps[0], maps[2]]), T0 + 40, T0 + 40.001, scaler, ox, oy)
            frame, centres = frames[k]
            assert float(np.abs(w2 - planted.expected(centres)).max()) > 8.0 / (368.0 / max(frame.shape[:2]))
    for hh in ranks:
        hh.close()


# ------------------------------------------------------------------------------------------ pyramid sharding
def test_pyramid_shards_reassemble(weights, oracle_net):
    """configs[3] without a second GPU: three rank-handles (one scale each) run their own pre-processing and conv
    stack; stacking their maps (what ncclAllGather delivers) and post-processing equals the unsharded result."""
    import oracle
    from tests import helpers
    frame = helpers.synth_frame(4242, 400, 360, smooth=True)
    n = _native()
    full = _handle(BASELINE_SCALES, weights)
    fb, scaler, (ox, oy) = full.preprocess(frame)
    fmaps = full.forward(fb)
    ranks = [n.Handle(BASELINE_SCALES, pyramid=(r, 3)) for r in range(3)]
    gathered = []
    for r, h in enumerate(ranks):
        h.set_weights(weights)
        h.finalize()
        b, s, off = h.preprocess(frame)
        assert b.shape == (1, 368, 368, 3) and s == scaler and off == [ox, oy]
        assert np.array_equal(b[0], fb[r])                      # rank r builds scale r of the pyramid, bit for bit
        m = h.forward(b)
        # the S images are independent through the net; a 1-image launch plan may split K differently (other
        # summation order), so equality is to fp32 rounding, not bitwise
        assert np.abs(m[0] - fmaps[r]).max() <= 1e-5 * np.abs(fmaps[r]).max()
        gathered.append(m[0])
        with pytest.raises(n.VnectError):                       # no communicator yet: inference must refuse, not hang
            h.infer(frame, T0, T0)
    gathered = np.stack(gathered)
    j2, j3 = ranks[0].postprocess(gathered, T0, T0 + 0.001, scaler, ox, oy)
    ref = oracle.OracleEstimator(scales=BASELINE_SCALES)
    o2, o3 = ref.postprocess(gathered, T0, T0 + 0.001, scaler, ox, oy)
    assert np.array_equal(j2, o2) and np.array_equal(j3, o3)    # a sharded rank's post-processing == oracle on the gathered maps
    # sharded vs unsharded conv stack differ by fp32 rounding only (checked per rank above); their post-processing is
    # exact arithmetic, so the unsharded handle fed the GATHERED maps must return the sharded result bit for bit
    f2, f3 = full.postprocess(gathered, T0, T0 + 0.001, scaler, ox, oy)
    assert np.array_equal(j2, f2) and np.array_equal(j3, f3)
    for h in ranks + [full]:
        h.close()


@pytest.mark.parametrize("two_launches", [False, True])
def test_pyramid_p2p_missing_peer_fails_the_frame(weights, monkeypatch, two_launches):
    """A rank whose peers never show up must get VNECT_E_COMM from the frame after the bounded wait -- never a hang.  Both forms of
    the post-processing honour the failed-exchange word (post_kernel, and joints_kernel behind VNECT_NO_POST_MERGE=1): the frame's
    joints stage is skipped on the device, so the filter banks do not advance on stale maps."""
    from tests import helpers
    n = _native()
    monkeypatch.setenv("VNECT_XCHG_SPINS", "20000")
    if two_launches:
        monkeypatch.setenv("VNECT_NO_POST_MERGE", "1")
    ranks = [n.Handle([1.0, 0.7], pyramid=(r, 2), exchange=n.XCHG_P2P) for r in range(2)]
    for h in ranks:
        h.set_weights(weights)
        h.finalize()
    blobs = [h.p2p_export() for h in ranks]
    for r, h in enumerate(ranks):
        h.p2p_init(r, 2, blobs)
    with pytest.raises(n.VnectError) as e:
        ranks[0].infer(helpers.synth_frame(3), T0, T0)     # rank 1 never submits this frame
    assert e.value.code == n.E_COMM
    for h in ranks:
        h.close()


P2P_WORKER = r"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, %r)
from vnect_amd import _native
from vnect_amd.weights import synthetic_weights
from tests import helpers
rank, world, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
scales = [1.0, 0.8, 0.6]
h = _native.Handle(scales, pyramid=(rank, world), exchange=_native.XCHG_P2P)
h.set_weights(synthetic_weights())
h.finalize()
open(os.path.join(d, "blob%%d.tmp" %% rank), "wb").write(h.p2p_export())
os.rename(os.path.join(d, "blob%%d.tmp" %% rank), os.path.join(d, "blob%%d" %% rank))
t0 = time.time()
while not all(os.path.exists(os.path.join(d, "blob%%d" %% r)) for r in range(world)):
    assert time.time() - t0 < 120
    time.sleep(0.05)
h.p2p_init(rank, world, [open(os.path.join(d, "blob%%d" %% r), "rb").read() for r in range(world)])
out = []
for k in range(4):
    frame = helpers.synth_frame(8000 + k, smooth=True)
    j2, j3 = h.infer(frame, 1.7e9 + k / 30, 1.7e9 + k / 30 + 0.001)
    out.append([j2.tolist(), j3.astype(np.float64).tolist()])
print(json.dumps(out), flush=True)
h.close()
"""


def test_pyramid_p2p_across_processes(weights, tmp_path):
    """The same exchange with one PROCESS per rank (the deployment shape: one process per GPU), all three on this box's one GPU:
    the exchange blocks are IPC-mapped (hipIpcGetMemHandle / hipIpcOpenMemHandle), each rank waits in-kernel for the other
    processes' stores.  All ranks must print the same joints, equal to the one-process sharded result (conv stack of one image
    per rank, so compared against sharded handles here, not against the 3-image batch whose K split may differ)."""
    import subprocess
    import sys
    from tests import helpers
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "p2p_worker.py"
    script.write_text(P2P_WORKER % root)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "3", str(tmp_path)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(3)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    assert outs[0] == outs[1] == outs[2]
    # reference: the same three one-image conv stacks run one after the other in THIS process (rank handles without an exchange:
    # pre-processing + vnect_forward), their maps stacked on the host, and one handle's post-processing over the stack with
    # its filter chain in lockstep -- bit for bit what the exchanged frames must give.  (Three rank handles of ONE process on
    # ONE device cannot wait for each other in-kernel: their streams may share a hardware queue, so a waiting kernel can sit
    # in front of the kernel it waits for.  One process per GPU -- the deployment shape -- has a queue of its own.)
    n = _native()
    ranks = [n.Handle(BASELINE_SCALES, pyramid=(r, 3)) for r in range(3)]
    for h in ranks:
        h.set_weights(weights)
        h.finalize()
    for k in range(4):
        frame = helpers.synth_frame(8000 + k, smooth=True)
        maps = []
        for h in ranks:
            b, scaler, (ox, oy) = h.preprocess(frame)
            maps.append(h.forward(b)[0])
        j2, j3 = ranks[0].postprocess(np.stack(maps), 1.7e9 + k / 30, 1.7e9 + k / 30 + 0.001, scaler, ox, oy)
        assert np.array_equal(np.array(outs[0][k][0]), j2), k
        assert np.array_equal(np.array(outs[0][k][1]).astype(np.float32), j3), k
    for h in ranks:
        h.close()


def test_pyramid_rccl_single_rank(weights):
    """The RCCL plumbing itself (ncclCommInitRank + ncclAllGather on the handle's stream) with a 1-rank communicator:
    a 1-scale sharded handle must return exactly what the plain 1-scale handle returns."""
    from tests import helpers
    n = _native()
    frame = helpers.synth_frame(77, smooth=True)
    plain = _handle([1.0], weights)
    shard = n.Handle([1.0], pyramid=(0, 1))
    shard.set_weights(weights)
    shard.finalize()
    shard.comm_init(0, 1, n.Handle.comm_unique_id())
    for k in range(3):
        t = T0 + k / 30
        a2, a3 = plain.infer(frame, t, t + 0.001)
        b2, b3 = shard.infer(frame, t, t + 0.001)
        assert np.array_equal(a2, b2) and np.array_equal(a3, b3), k
    plain.close(), shard.close()


# ------------------------------------------------------------------------------------------ bf16 path (configs[2])
def test_bf16_path_gated_against_fp32(weights, oracle_net, h3):
    """bf16 MFMA conv path: activations and weights are bf16 (8 significant bits), accumulation fp32, final maps
    and post-processing fp32/f64.  Gate (calibrated on MI355X, seeded synthetic weights, 54 layers deep):
    every layer <= 4e-2 * max|ref|, final maps <= 3e-2 * max|ref|; the fp32 path sits at 3e-6 on the same input."""
    import oracle
    from tests import helpers
    batch, _, _ = oracle.gen_input_batch(helpers.synth_frame(1234, smooth=True), BASELINE_SCALES)
    ref = oracle_net.forward(batch)
    hb = _handle(BASELINE_SCALES, weights, precision=_native().BF16, keep_activations=True)
    out = hb.forward(batch)
    f32 = h3.forward(batch)
    rows = []
    for n in ["conv1", "pool1", "res2a", "res2c", "res3d", "res4a_branch2b", "res4f", "res5a", "res5b_branch2c_new",
              "res5c_branch2a_feat", "res5c_branch2b", "res5c_branch2c"]:
        a, r = hb.activation(n), oracle_net.activation(n)
        assert a.shape == r.shape, n
        rows.append((n, float(np.abs(a - r).max() / np.abs(r).max())))
    _log("bf16_layer_errors.json", rows)
    for n, e in rows:
        print("%-24s rel err %.3g" % (n, e))
    e_out = float(np.abs(out - ref).max() / np.abs(ref).max())
    e_f32 = float(np.abs(f32 - ref).max() / np.abs(ref).max())
    print("final maps: bf16 %.3g, fp32 %.3g" % (e_out, e_f32))
    assert all(e <= 4e-2 for _, e in rows), rows
    assert e_out <= 3e-2 and e_f32 <= 1e-4
    # the preprocessing is the same integer arithmetic, rounded once to bf16 at the end
    frame = helpers.synth_frame(9, 300, 420, smooth=True)
    bb, s, off = hb.preprocess(frame)
    rb, rs, roff = oracle.gen_input_batch(frame, BASELINE_SCALES)
    assert s == rs and off == roff and np.abs(bb - rb).max() <= 2 ** -8
    # end to end: joints from planted-peak-free noise maps are tie-prone; require finite, well-formed output and
    # agreement of most joints within one heat-map cell (8 px) with the fp32 path
    j2b, j3b = hb.infer(frame, T0, T0 + 0.001)
    h3.reset_filters()
    j2f, j3f = h3.infer(frame, T0, T0 + 0.001)
    assert np.all(np.isfinite(j2b)) and np.all(np.isfinite(j3b))
    close = np.all(np.abs(j2b - j2f) <= 8.0 / min(s, 1.0) + 1e-9, axis=1)
    same = np.all(j2b == j2f, axis=1)
    mb, mf = hb.activation("res5c_branch2c"), h3.activation("res5c_branch2c")
    # 3-D read-off where both paths sit on the same pixel: the location maps differ by <= 3e-2 * max|map| (gate above), the
    # read-off is a convex blend of 4 cells x 100 (mm), and the root joint's row is subtracted: 2 * 3e-2 * max|xyz maps| * 100
    bound3 = 2 * 3e-2 * float(np.abs(mf[..., 21:]).max()) * 100
    d3 = np.abs(j3b - j3f)
    print("bf16 vs fp32 joints: %d/21 within one cell, %d/21 on the same pixel; max 3-D diff on those %.3g mm (bound %.3g)"
          % (close.sum(), same.sum(), float(d3[same].max()) if same.any() else -1, bound3))
    # 2-D: heat-maps of random weights are noise-like, so bf16 noise may move an arg-max to another near-maximal cell.  What
    # MUST hold given |bf16 maps - fp32 maps| <= eps everywhere (gate above; the merge and the x8 upsample are convex blends):
    # the fp32 heat-map at the bf16 arg-max is within 2 eps of its own maximum -- for every joint.
    eps = 3e-2 * float(np.abs(mf).max())
    avg_f = oracle.merge_scales(mf, BASELINE_SCALES)[0]
    raw_b = oracle.extract_2d(oracle.merge_scales(mb, BASELINE_SCALES)[0])
    for j in range(21):
        up = oracle.resize(np.ascontiguousarray(avg_f[:, :, j]), 8.0)
        assert up[int(raw_b[j, 0]), int(raw_b[j, 1])] >= up.max() - 2 * eps, j
    # (no "most joints within one cell" floor here: where a heat-map HAS a maximum -- a margin over every other cell that bf16 noise
    # cannot bridge -- the bf16 arg-max must sit in the fp32 cell, for every such joint: test_bf16_margin_conditioned_joints)
    if same[14]:
        assert np.all(d3[same] <= bound3)
    assert float(np.abs(mb - mf).max()) <= 3e-2 * float(np.abs(mf).max())
    hb.close()


def _round_bf16(a):
    """float32 -> nearest-even bfloat16 -> float32 (what the bf16 path's per-layer rounding does to a value)."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(a))


def test_bf16_benchmarked_plan_parity(weights, oracle_net, h3):
    """The bf16 plan bench.py times is the ARENA plan (tail GEMM + bone fusion on, S = 3), not the private-buffer one the
    per-layer gate reads back.  (1) its maps equal the private-buffer handle's bit for bit (the fp32 twin of this check is in
    test_conv_stack_every_layer) and sit within 3e-2 of the oracle; (2) over 4 frames incl. non-square ones, for EVERY joint the
    fp32 heat-map at the bf16 arg-max is within 2 eps of its maximum, and the arena handle's joints equal the private-buffer
    handle's; (3) where the heat-maps have a real maximum (planted peaks, utils.py:153-175 semantics) bf16-rounded maps give
    21/21 joints within one heat-map cell."""
    import oracle
    from tests import helpers
    n = _native()
    batch, _, _ = oracle.gen_input_batch(helpers.synth_frame(1234, smooth=True), BASELINE_SCALES)
    ref = oracle_net.forward(batch)
    fused = _handle(BASELINE_SCALES, weights, precision=n.BF16)                              # what bench.py builds
    plain = _handle(BASELINE_SCALES, weights, precision=n.BF16, keep_activations=True)       # what the per-layer gate reads
    names = [L["name"] for L in fused.layers()]
    assert any(">" in x for x in names) and any("bone_length" in x and "deconv" in x for x in names), names   # fused launches present
    assert not any(">" in L["name"] for L in plain.layers())
    mf, mp = fused.forward(batch), plain.forward(batch)
    assert np.array_equal(mf, mp)                                                            # (1)
    assert float(np.abs(mf - ref).max() / np.abs(ref).max()) <= 3e-2
    worst_close = 21
    for k, (H, W) in enumerate([(368, 368), (538, 368), (240, 320), (368, 368)]):           # (2)
        frame = helpers.synth_frame(777 + k, H, W, smooth=True)
        t = T0 + 500 + k / 30
        j2a, j3a = fused.infer(frame, t, t + 0.001)
        j2p, j3p = plain.infer(frame, t, t + 0.001)
        assert np.array_equal(j2a, j2p) and np.array_equal(j3a, j3p), k
        if k == 0:
            h3.reset_filters()
        j2f, j3f = h3.infer(frame, t, t + 0.001)
        mb, m32 = fused.activation("res5c_branch2c"), h3.activation("res5c_branch2c")
        top = float(np.abs(m32).max())
        assert float(np.abs(mb - m32).max()) <= 3e-2 * top, k
        eps = 3e-2 * top
        avg_f = oracle.merge_scales(m32, BASELINE_SCALES)[0]
        raw_b = oracle.extract_2d(oracle.merge_scales(mb, BASELINE_SCALES)[0])
        for j in range(21):
            up = oracle.resize(np.ascontiguousarray(avg_f[:, :, j]), 8.0)
            assert up[int(raw_b[j, 0]), int(raw_b[j, 1])] >= up.max() - 2 * eps, (k, j)
        if k == 0:  # first frame: the filters are the identity, so joints_2d ARE the arg-max positions
            scaler = 368.0 / max(H, W)
            worst_close = min(worst_close, int(np.all(np.abs(j2a - j2f) <= 8.0 / scaler + 1e-9, axis=1).sum()))
    print("bf16 arena plan: worst frame has %d/21 joints within one cell of fp32 (noise heat-maps; informational -- the gate is the "
          "2-eps rule above and test_bf16_margin_conditioned_joints)" % worst_close)
    # (3) planted peaks: the post-processing of bf16-rounded maps against the fp32 maps
    for seed in (5, 6, 7):
        maps = helpers.synth_maps(seed, 3)
        fused.reset_filters()
        a2, a3 = fused.postprocess(_round_bf16(maps), T0, T0 + 0.001)
        fused.reset_filters()
        b2, b3 = fused.postprocess(maps, T0, T0 + 0.001)
        assert np.all(np.abs(a2 - b2) <= 8.0), seed       # 21/21 within one heat-map cell (oracle on the same maps: 0, 1 and 7 px)
    fused.close(), plain.close()


def test_bf16_margin_conditioned_joints(weights):
    """What the bf16 path owes the joints, THROUGH the bf16 net (utils.py:153-219 semantics).  Heat-maps of random weights are noise-like: an
    arg-max may legally jump between near-equal cells (the 2-eps rule of the two tests above), and measured on them NO maximum clears the
    rest of its map by 2 eps (tools/bf16_margin_probe.py: the largest margin of 252 joints is 1.7 eps).  So this test runs weights whose
    heat-maps HAVE maxima -- tests/planted.py: heat-map j peaks ON a blob painted into the frame, over the random net's own noise floor;
    the oracle finds those joints to the pixel, tests/test_planted.py -- through the benchmarked bf16 arena plan and the fp32 plan, 16
    frames (square, the test picture's 538x368, landscape, portrait), every joint:

      * known answer: the fp32 AND the bf16 joints_2d lie on the planted blob, within one heat-map cell (fp32: within one box pixel);
      * map gate: |bf16 maps - fp32 maps| <= eps = 3e-2 * max|fp32 maps| (every frame);
      * the 2-eps rule for every joint; and margin-conditioned exactness: every (frame, joint) whose fp32 maximum beats the best value
        outside its own heat-map cell by more than 2 eps -- merge and x8 upsample are convex blends, bf16 noise cannot bridge that --
        has its bf16 arg-max IN THAT CELL: all of them, and at least 100 such pairs must exist;
      * joints_3d, every joint of every frame: the bf16 handle's read-off against the FP32 location maps read at the same (bf16) pixels
        -- root row included -- within the bound the map gate implies (2 eps x 100 mm, times the read-off's extrapolation weights at the
        borders); and where the pixel AND the root's pixel equal the fp32 path's, against the fp32 path's joints_3d.
    No floor of the kind "most joints within one cell" is left anywhere."""
    import oracle
    from tests import planted
    from tests.test_planted import cell_margin
    n = _native()
    pw = planted.weights()
    hb = _handle(BASELINE_SCALES, pw, precision=n.BF16)         # the arena plan bench.py times
    hf = _handle(BASELINE_SCALES, pw)
    assert any(">" in L["name"] for L in hb.layers())           # fused launches (tail / chain GEMMs) are in this plan
    shapes = [(368, 368), (538, 368), (240, 320), (368, 300)]
    pairs, held, below, rows = 0, 0, 0, []
    worst3, map_err, off_f, off_b = 0.0, 0.0, 0.0, 0.0
    for k in range(16):
        H, W = shapes[k % 4]
        frame, centres = planted.frame(300 + k, H, W)
        want = planted.expected(centres)
        scaler = 368.0 / max(H, W)
        t = T0 + 900 + k
        hb.reset_filters(), hf.reset_filters()                  # first frame of a stream: the filters are the identity
        j2b, j3b = hb.infer(frame, t, t + 0.001)
        mb = hb.activation("res5c_branch2c")
        j2f, j3f = hf.infer(frame, t, t + 0.001)
        mf = hf.activation("res5c_branch2c")
        off_f, off_b = max(off_f, float(np.abs(j2f - want).max()) * scaler), max(off_b, float(np.abs(j2b - want).max()) * scaler)
        assert np.abs(j2f - want).max() <= 1.0 / scaler, (k, np.abs(j2f - want).max())           # the known answer, fp32
        assert np.abs(j2b - want).max() <= 8.0 / scaler, (k, np.abs(j2b - want).max())           # ... and bf16: the same cell or its neighbour
        top = float(np.abs(mf).max())
        eps = 3e-2 * top
        map_err = max(map_err, float(np.abs(mb - mf).max()) / top)
        assert float(np.abs(mb - mf).max()) <= eps, (k, float(np.abs(mb - mf).max()) / top)
        avg_f, avg_b = oracle.merge_scales(mf, BASELINE_SCALES), oracle.merge_scales(mb, BASELINE_SCALES)
        raw_f, raw_b = oracle.extract_2d(avg_f[0]), oracle.extract_2d(avg_b[0])
        for j in range(21):
            up = oracle.resize(np.ascontiguousarray(avg_f[0][:, :, j]), 8.0)
            assert up[int(raw_b[j, 0]), int(raw_b[j, 1])] >= up.max() - 2 * eps, (k, j)          # the 2-eps rule, every joint
            gap = cell_margin(up, raw_f[j])
            same_cell = bool(np.all(raw_b[j] // 8 == raw_f[j] // 8))
            rows.append((k, j, round(gap / eps, 3), same_cell))
            if gap > 2 * eps:
                pairs += 1
                held += same_cell
                assert same_cell, "frame %d joint %d: margin %.2f eps, bf16 arg-max %s left the fp32 cell of %s" % (k, j, gap / eps, raw_b[j], raw_f[j])
            else:
                below += 1
        # joints_3d: the bf16 read-off against the fp32 location maps at the bf16 pixels (first frame: unfiltered positions)
        at_b = oracle.extract_3d(raw_b, avg_f[1], avg_f[2], avg_f[3])
        own = oracle.extract_3d(raw_b, avg_b[1], avg_b[2], avg_b[3])
        assert np.array_equal(own, j3b), k           # the GPU's post-processing of its own maps is exact arithmetic
        # read-off weights: convex inside; a pixel left of 3.5 extrapolates with weights (1 + a, -a), a <= 7/16, per axis
        amp = 1.0 + 2 * (7.0 / 16)
        bound3 = 2 * (eps * 100) * amp * amp
        d3 = float(np.abs(j3b.astype(np.float64) - at_b).max())
        worst3 = max(worst3, d3 / bound3)
        assert d3 <= bound3, (k, d3, bound3)
        same_px = np.all(raw_b == raw_f, axis=1)
        if same_px[14]:
            assert np.all(np.abs(j3b - j3f)[same_px] <= bound3), k
    _log("bf16_margin_pairs.json", {"pairs_with_margin_over_2eps": pairs, "of_them_in_the_fp32_cell": held, "pairs_below": below,
                                    "bf16_map_err_over_max": map_err, "worst_3d_over_bound": worst3, "fp32_offset_from_blob_box_px": off_f,
                                    "bf16_offset_from_blob_box_px": off_b, "rows": rows})
    print("bf16 margin gate: %d of %d (frame, joint) pairs have a margin > 2 eps, all %d in the fp32 cell; bf16 map error %.3g of max (gate 3e-2); "
          "joints vs the planted blobs: fp32 <= %.2f px, bf16 <= %.2f px; 3-D read-off at most %.2f of its bound"
          % (pairs, 16 * 21, held, map_err, off_f, off_b, worst3))
    assert pairs >= 100, "only %d pairs with a real maximum" % pairs
    hb.close(), hf.close()


# ------------------------------------------------------------------------------------------ split-product fp32 path
def test_split_product_path_meets_the_fp32_gates(weights, oracle_net):
    """precision = FP32_SPLIT: fp32 tensors and accumulators, but the 64x64-tile layers form their products on the bf16 matrix pipe from
    exact three-way splits of both operands (6 of the 9 piece products; conv.hip, X3).  It is an fp32-class path, so it is held to
    the FP32 gates, not the bf16 ones: every layer <= 1e-4 * max|oracle| (its error is printed beside the fp32 instruction's), arena
    plan == private-buffer plan bit for bit, and the whole __call__ through the every-frame, every-joint gate of _EndToEnd."""
    import oracle
    from tests import helpers
    n = _native()
    batch, _, _ = oracle.gen_input_batch(helpers.synth_frame(1234, smooth=True), BASELINE_SCALES)
    ref = oracle_net.forward(batch)
    hs = _handle(BASELINE_SCALES, weights, precision=n.FP32_SPLIT, keep_activations=True)
    hf = _handle(BASELINE_SCALES, weights, keep_activations=True)
    out_s, out_f = hs.forward(batch), hf.forward(batch)
    rows = []
    for name in ["pool1", "res2a_branch2a", "res2a", "res2c", "res3a", "res3d", "res4a_branch2b", "res4f", "res5a_branch2a_new", "res5a",
                 "res5b_branch2c_new", "res5c_branch2a_feat", "res5c_branch2b", "res5c_branch2c"]:
        r = oracle_net.activation(name)
        top = float(np.abs(r).max())
        es, ef = float(np.abs(hs.activation(name) - r).max()) / top, float(np.abs(hf.activation(name) - r).max()) / top
        rows.append((name, es, ef))
        print("%-24s split %.3g   fp32 instruction %.3g" % (name, es, ef))
    _log("split_layer_errors.json", rows)
    assert all(es <= 1e-4 for _, es, _ in rows), rows
    assert float(np.abs(out_s - ref).max()) <= 1e-4 * float(np.abs(ref).max())
    hs.close(), hf.close()
    arena = _handle(BASELINE_SCALES, weights, precision=n.FP32_SPLIT)
    assert any(">" in L["name"] for L in arena.layers())           # the tail-fused launches are in this plan
    # (the arena plan's tail GEMMs multiply with the fp32 instruction, the private-buffer plan's stand-alone 1x1 layers by splits: equal
    # to fp32 rounding, not bit for bit -- in this mode the last bits depend on the launch plan, like any change of summation order)
    assert float(np.abs(arena.forward(batch) - out_s).max()) <= 2e-5 * float(np.abs(ref).max())
    e2e = _EndToEnd(BASELINE_SCALES, oracle_net)
    for k, (H, W) in enumerate([(368, 368), (538, 368), (240, 320), (368, 368)]):
        frame = helpers.synth_frame(8800 + k, H, W, smooth=True)
        t = T0 + 300 + k / 30
        j2, j3 = arena.infer(frame, t, t + 0.001)
        e2e.check(frame, t, t + 0.001, j2, j3, arena.activation("res5c_branch2c"), (H, W))
    print("split-product path: legal arg-max ties %d, worst 3-D excess over tolerance %.3g" % (e2e.ties, e2e.worst3))
    arena.close()


def test_split_product_path_vs_float64(weights):
    """Is the split-product path "reduced precision"?  Both GPU paths against the TRUE result: the torch float64 restatement of the
    graph (tests/torch_net.py, independent of the C oracle) on one image.  The split-product maps must be as close to float64 as the
    fp32 instruction's are (within 1.5x of its max error and of its RMS error), and both far inside the fp32 gate."""
    import oracle
    import torch
    from tests import helpers, torch_net
    n = _native()
    torch.set_num_threads(16)
    batch, _, _ = oracle.gen_input_batch(helpers.synth_frame(4242, smooth=True), [1.0])
    with torch.inference_mode():
        ref = torch_net.forward(weights, batch, dtype=torch.float64).numpy()
    top = float(np.abs(ref).max())
    errs = {}
    for name, prec in (("fp32 instruction", n.FP32), ("split product", n.FP32_SPLIT)):
        h = _handle([1.0], weights, precision=prec)
        d = h.forward(batch).astype(np.float64) - ref
        h.close()
        errs[name] = (float(np.abs(d).max()) / top, float(np.sqrt((d * d).mean())) / top)
        print("%-18s vs float64: max %.3g  rms %.3g (of max|map|)" % (name, *errs[name]))
    f, s = errs["fp32 instruction"], errs["split product"]
    assert s[0] <= 1.5 * f[0] and s[1] <= 1.5 * f[1], errs
    assert s[0] <= 2e-5 and f[0] <= 2e-5
