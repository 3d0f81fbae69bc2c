"""CPU: the product's host-side planning code (vnect_amd/csrc/hostplan.h -- resize / merge / upsample tables, weight packing incl. the
transposed conv's phase layout, the activation arena's first fit, the tile choice, the fused stem's row groups) behind its C shim
(hostplan_capi.cpp), checked against the oracle's resizes and against index formulas written down here independently -- and the
same tests once more under AddressSanitizer + UndefinedBehaviorSanitizer (test_hostplan_clean_under_asan_ubsan).  The code under
test is what the host runtime (rt_plan.cpp) compiles into libvnect_hip.so; the shim libraries are test infrastructure and never loaded by the product."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vnect_amd", "csrc")
SO = os.environ.get("VNECT_HOSTPLAN_SO") or os.path.join(ROOT, "vnect_amd", "lib", "libvnect_hostplan.so")

f32p, f64p, u8p, i32p, u64p = (C.POINTER(t) for t in (C.c_float, C.c_double, C.c_uint8, C.c_int, C.c_uint64))


def _p(a, t):
    return a.ctypes.data_as(t)


@pytest.fixture(scope="module")
def hp():
    if "VNECT_HOSTPLAN_SO" not in os.environ:
        subprocess.check_call(["make", "-C", CSRC, "hostplan"], stdout=subprocess.DEVNULL)
    L = C.CDLL(SO)
    L.hp_resize_u8.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_double, u8p, i32p, i32p]
    L.hp_gen_input_batch.argtypes = [u8p, C.c_int, C.c_int, C.c_int64, f64p, C.c_int, f32p, f64p, i32p, i32p, C.c_char_p, C.c_int]
    L.hp_merge.argtypes = [f32p, f64p, C.c_int, f64p]
    L.hp_merge_geo.argtypes = [f32p, f64p, C.c_int, f64p]
    L.hp_axis_mismatches.argtypes = [C.c_int, C.c_int, C.c_double]
    L.hp_extract_2d.argtypes = [f64p, C.c_int, f64p]
    L.hp_squarify.argtypes = [C.c_int, C.c_int, f64p, i32p, i32p, i32p, i32p, i32p, C.c_char_p, C.c_int]
    L.hp_pack_conv.argtypes = [f32p] + [C.c_int] * 9 + [f32p]
    L.hp_pack_tail.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p]
    L.hp_pack_deconv.argtypes = [f32p, f32p, C.c_int, C.c_int, f32p, i32p, i32p]
    L.hp_fold_bn.argtypes = [f32p] * 4 + [C.c_int, C.c_int] + [f32p] * 3
    L.hp_to_bf16.argtypes = [C.c_float]
    L.hp_to_bf16.restype = C.c_uint16
    L.hp_from_bf16.argtypes = [C.c_uint16]
    L.hp_from_bf16.restype = C.c_float
    L.hp_pack_split3.argtypes = [f32p, C.c_int, C.c_int, C.POINTER(C.c_uint16)]
    L.hp_arena.argtypes = [i32p, i32p, u64p, C.c_int, u64p]
    L.hp_arena.restype = C.c_uint64
    L.hp_choose_tile.argtypes = [C.c_int] * 7 + [C.c_char_p] * 3 + [i32p]
    L.hp_choose_tile96.argtypes = [C.c_int] * 7 + [C.c_char_p] * 3 + [i32p]
    L.hp_stem_groups.argtypes = [C.c_int, u8p]
    L.hp_stem_frame_fits.argtypes = [f64p, C.c_int, C.c_int, C.c_int]
    return L


# ------------------------------------------------------------------------------------------ tables vs the oracle's resizes
def test_u8_tables_reproduce_the_oracle_resize(hp):
    """build_u8_tab + the fixed-point blend (what pyramid.h evaluates on the device) against oracle.resize on uint8 images: the
    pyramid factors of both scale sets, squarify factors of odd frame shapes, and the single-tap right border."""
    import oracle
    from tests import helpers
    cases = [(368, 368, s) for s in (0.8, 0.6, 0.85, 0.7, 0.5, 0.9999)]
    cases += [(h, w, 368.0 / max(h, w)) for h, w in ((538, 368), (240, 320), (200, 120), (77, 368), (368, 91), (1080, 1920), (33, 47))]
    for k, (h, w, f) in enumerate(cases):
        img = np.ascontiguousarray(helpers.synth_frame(400 + k, h, w, smooth=(k % 2 == 0)))   # (the smooth frames are strided views)
        want = oracle.resize(img, f)
        dh, dw = C.c_int(), C.c_int()
        assert hp.hp_resize_u8(_p(img, u8p), h, w, 3, f, None, C.byref(dh), C.byref(dw)) == 0
        assert (dh.value, dw.value) == want.shape[:2], (h, w, f)
        got = np.empty((dh.value, dw.value, 3), np.uint8)
        hp.hp_resize_u8(_p(img, u8p), h, w, 3, f, _p(got, u8p), C.byref(dh), C.byref(dw))
        assert np.array_equal(got, want), (h, w, f)
    # hostile factors are refused, not overflowed
    for f in (0.0, -1.0, float("nan"), 1e30, 1e-30):
        assert hp.hp_resize_u8(_p(np.zeros((8, 8, 3), np.uint8), u8p), 8, 8, 3, f, None, C.byref(dh), C.byref(dw)) == -1


@pytest.mark.parametrize("shape", [(368, 368), (538, 368), (240, 320), (368, 200), (123, 368)])
def test_host_pyramid_equals_oracle_gen_input_batch(hp, shape):
    """squarify geometry + square-on-demand + per-scale tables + centre padding + `/255 - 0.4`, composed as the device composes
    them, against oracle.gen_input_batch (estimator.py:70-81): batch, scaler and offsets, bit for bit."""
    import oracle
    from tests import helpers
    frame = np.ascontiguousarray(helpers.synth_frame(510 + shape[0], shape[0], shape[1], smooth=True))
    for scales in ([1.0, 0.8, 0.6], [1, 0.85, 0.7]):
        want, ws, woff = oracle.gen_input_batch(frame, scales)
        got = np.empty((len(scales), 368, 368, 3), np.float32)
        sc, ox, oy = C.c_double(), C.c_int(), C.c_int()
        err = C.create_string_buffer(200)
        s64 = np.array(scales, np.float64)
        rc = hp.hp_gen_input_batch(_p(frame, u8p), shape[0], shape[1], frame.strides[0], _p(s64, f64p), len(scales), _p(got, f32p),
                                   C.byref(sc), C.byref(ox), C.byref(oy), err, 200)
        assert rc == 0, err.value
        assert sc.value == ws and [ox.value, oy.value] == list(woff)
        assert np.array_equal(got, want), (shape, scales)


def test_squarify_rejects_what_the_reference_rejects(hp):
    out = [C.c_double(), C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()]
    refs = [C.byref(x) for x in out]
    err = C.create_string_buffer(200)
    assert hp.hp_squarify(368, 368, *refs, err, 200) == 0 and out[5].value == 1 and out[0].value == 1.0   # a copy
    assert hp.hp_squarify(538, 368, *refs, err, 200) == 0 and out[5].value == 0 and out[3].value == 368
    for h, w in ((0, 5), (5, 0), (9000, 10), (-3, 7)):
        assert hp.hp_squarify(h, w, *refs, err, 200) == -1 and err.value


def test_merge_and_upsample_tables_vs_oracle(hp):
    """build_merge_tab applied like post.hip's merged_cell vs oracle.merge_scales (estimator.py:105-129), and build_up_tab applied like
    the arg-max kernel vs oracle.extract_2d (utils.py:153-175)."""
    import oracle
    from tests import helpers
    for scales in ([1.0, 0.8, 0.6], [1, 0.85, 0.7], [1.0]):
        maps = helpers.synth_maps(77, len(scales))
        want = oracle.merge_scales(maps, scales)              # 4 x (46,46,21) f64
        got = np.empty((46, 46, 84), np.float64)
        s64 = np.array(scales, np.float64)
        assert hp.hp_merge(_p(maps, f32p), _p(s64, f64p), len(scales), _p(got, f64p)) == 0
        for q in range(4):
            assert np.array_equal(got[:, :, 21 * q:21 * q + 21], want[q]), (scales, q)
        heat = np.ascontiguousarray(want[0])
        j = np.empty((21, 2), np.float64)
        assert hp.hp_extract_2d(_p(heat, f64p), 21, _p(j, f64p)) == 0
        assert np.array_equal(j, oracle.extract_2d(heat))


def test_per_entry_axis_functions_equal_the_tables(hp):
    """axis.h (what post.hip evaluates on the device since round 3: every tap and weight computed where it is used) against the
    whole-table builders, entry by entry and bit by bit: the merge's resizes over a sweep of scales, the x8 upsample, and shapes with
    the far-border single tap; then the merge evaluated without tables (MergeGeo, as the kernels get it) against the oracle."""
    import oracle
    from tests import helpers
    for s in list(np.linspace(0.3, 1.0, 71)) + [0.85, 0.7, 0.6, 0.8, 1 / 3, 0.9999]:
        f = 1.0 / s
        ds = int(np.rint(46 * f))  # cv_round: round half to even, like np.rint
        assert hp.hp_axis_mismatches(46, ds, 1.0 / f) == 0, s
    assert hp.hp_axis_mismatches(46, 368, 1.0 / 8.0) == 0
    for ssize, dsize in ((5, 17), (368, 294), (7, 7), (2, 9), (1, 4), (100, 3)):
        assert hp.hp_axis_mismatches(ssize, dsize, ssize / dsize) == 0, (ssize, dsize)
    for scales in ([1.0, 0.8, 0.6], [1, 0.85, 0.7], [1.0], [0.9, 0.75, 0.5, 0.45, 1.0]):
        maps = helpers.synth_maps(78, len(scales))
        want = oracle.merge_scales(maps, scales)
        got = np.empty((46, 46, 84), np.float64)
        s64 = np.array(scales, np.float64)
        assert hp.hp_merge_geo(_p(maps, f32p), _p(s64, f64p), len(scales), _p(got, f64p)) == 0
        for q in range(4):
            assert np.array_equal(got[:, :, 21 * q:21 * q + 21], want[q]), (scales, q)


# ------------------------------------------------------------------------------------------ weight packing
def _pack(hp, W, k, cin, cout, cp, conv1, bf16, Npad, K, n0=0):
    out = np.zeros((Npad, K), np.float32)
    Wc = np.ascontiguousarray(W, np.float32)
    hp.hp_pack_conv(_p(Wc, f32p), k, cin, cout, cp, int(conv1), int(bf16), Npad, K, n0, _p(out, f32p))
    return out


def test_conv_weight_packing_round_trip(hp):
    """pack_conv for ordinary layers (k = (tap, channel), channels padded to the input's pixel stride), for conv1 in both precisions
    (rows of 8 NHWC4 pixels; bf16: row pairs) and for a paired launch (two layers concatenated along N): every weight lands exactly
    where the index formula says, everything else is zero."""
    rng = np.random.RandomState(3)
    # ordinary 3x3, cin 20 padded to a 32-channel pixel stride
    W = rng.randn(3, 3, 20, 7).astype(np.float32)
    P = _pack(hp, W, 3, 20, 7, 32, False, False, 32, 9 * 32)
    ref = np.zeros_like(P)
    for ky in range(3):
        for kx in range(3):
            ref[:7, (ky * 3 + kx) * 32:(ky * 3 + kx) * 32 + 20] = W[ky, kx].T
    assert np.array_equal(P, ref)
    # conv1 fp32: [n][ky][8 px][4 ch]
    W1 = rng.randn(7, 7, 3, 64).astype(np.float32)
    P = _pack(hp, W1, 7, 3, 64, 32, True, False, 64, 224).reshape(64, 7, 8, 4)
    assert np.array_equal(P[:, :, :7, :3], W1.transpose(3, 0, 1, 2)) and not P[:, :, 7].any() and not P[:, :, :, 3].any()
    # conv1 bf16: [n][row pair][row in pair][8 px][4 ch], the eighth row is zero
    P = _pack(hp, W1, 7, 3, 64, 32, True, True, 64, 256).reshape(64, 8, 8, 4)
    assert np.array_equal(P[:, :7, :7, :3], W1.transpose(3, 0, 1, 2)) and not P[:, 7].any() and not P[:, :, 7].any() and not P[..., 3].any()
    # a pair: columns [0, 64) from layer a, [64, 64 + 24) from layer b
    Wa, Wb = rng.randn(1, 1, 32, 64).astype(np.float32), rng.randn(1, 1, 32, 24).astype(np.float32)
    P = np.zeros((128, 32), np.float32)
    hp.hp_pack_conv(_p(Wa, f32p), 1, 32, 64, 32, 0, 0, 128, 32, 0, _p(P, f32p))
    Q = np.zeros((128, 32), np.float32)
    hp.hp_pack_conv(_p(Wb, f32p), 1, 32, 24, 32, 0, 0, 128, 32, 64, _p(Q, f32p))
    assert np.array_equal(P[:64], Wa[0, 0].T) and not P[64:].any() and np.array_equal(Q[64:88], Wb[0, 0].T) and not Q[:64].any()
    # tail GEMMs: (1,1,mid,cout) -> the MFMA's B-fragment order [column block][group q][lane = 32 hh + column][e], element
    # k = UQ q + UH hh + e of column 32 cb + column (fp32: UQ, UH = 8, 4; bf16: 16, 8); columns padded to whole blocks with zeros
    # (the tails' 1x1 layers; the chain GEMMs' -- next block's branch2a, 512 -> 128 and 256 -> 64; the stem's pair, 64 -> 64 + 256)
    for mid, cout, bf16 in ((64, 256, 0), (64, 256, 1), (128, 512, 0), (128, 512, 1), (128, 84, 0), (128, 84, 1), (512, 128, 0), (512, 128, 1),
                            (256, 64, 1), (64, 320, 0), (64, 320, 1)):
        Wt = rng.randn(1, 1, mid, cout).astype(np.float32)
        UQ, UH = (16, 8) if bf16 else (8, 4)
        nb, NQ = (cout + 31) // 32, mid // UQ
        T = np.full((nb, NQ, 2, 32, UH), np.nan, np.float32)
        hp.hp_pack_tail(_p(Wt, f32p), mid, cout, bf16, _p(T, f32p))
        Wp = np.zeros((mid, nb * 32), np.float32)
        Wp[:, :cout] = Wt[0, 0]
        for cb in range(nb):
            for q in range(NQ):
                for hh in range(2):
                    k0 = UQ * q + UH * hh
                    assert np.array_equal(T[cb, q, hh], Wp[k0:k0 + UH, 32 * cb:32 * cb + 32].T), (mid, cout, bf16, cb, q, hh)


def test_deconv_phase_layout_matches_the_transposed_conv(hp):
    """pack_deconv's four sub-pixel phases against conv2d_transpose(4x4, stride 2, SAME) written out directly
    (out[2i-1+ky, 2j-1+kx, oc] += in[i,j,ic] * W[ky,kx,oc,ic], vnect_model.py:188-196): a random input run through the packed phases
    (tap offsets dy / dx, weight rows, zero padding outside the image) must equal the direct sum."""
    rng = np.random.RandomState(5)
    W1 = rng.randn(4, 4, 63, 256).astype(np.float32)
    W2 = rng.randn(4, 4, 128, 256).astype(np.float32)
    Npad, K = 192, 1024
    wp = np.zeros((4, Npad, K), np.float32)
    dy, dx = np.zeros(16, np.int32), np.zeros(16, np.int32)
    hp.hp_pack_deconv(_p(W1, f32p), _p(W2, f32p), Npad, K, _p(wp, f32p), _p(dy, i32p), _p(dx, i32p))
    assert not wp[:, 191:].any()
    Hs = 5
    x = rng.randn(Hs, Hs, 256).astype(np.float64)
    Wall = np.concatenate([W2, W1], axis=2).astype(np.float64)        # (4,4,191,256): columns 0..127 branch2a, 128..190 the deltas
    direct = np.zeros((2 * Hs, 2 * Hs, 191))
    for i in range(Hs):
        for j in range(Hs):
            for ky in range(4):
                for kx in range(4):
                    oy, ox = 2 * i - 1 + ky, 2 * j - 1 + kx
                    if 0 <= oy < 2 * Hs and 0 <= ox < 2 * Hs:
                        direct[oy, ox] += Wall[ky, kx] @ x[i, j]
    phased = np.zeros_like(direct)
    for z in range(4):
        py, px = z >> 1, z & 1
        for i in range(Hs):
            for j in range(Hs):
                acc = np.zeros(191)
                for t in range(4):
                    ii, jj = i + dy[z * 4 + t], j + dx[z * 4 + t]
                    if 0 <= ii < Hs and 0 <= jj < Hs:
                        acc += wp[z, :191, t * 256:(t + 1) * 256].astype(np.float64) @ x[ii, jj]
                phased[2 * i + py, 2 * j + px] = acc
    assert np.allclose(phased, direct, rtol=0, atol=1e-9)
    # BN folding: (acc + bias) * scale + shift == gamma * (acc - mean) / sqrt(var + 1e-3) + beta, in float32
    g, b, m, v = (rng.rand(128).astype(np.float32) + 0.5 for _ in range(4))
    bias, scale, shift = (np.zeros(192, np.float32) for _ in range(3))
    hp.hp_fold_bn(_p(g, f32p), _p(b, f32p), _p(m, f32p), _p(v, f32p), 128, 192, _p(bias, f32p), _p(scale, f32p), _p(shift, f32p))
    assert np.array_equal(bias[:128], -m) and np.array_equal(shift[:128], b)
    assert np.array_equal(scale[:128], g * (np.float32(1) / np.sqrt(v + np.float32(0.001), dtype=np.float32)))
    assert not bias[128:].any() and np.all(scale[128:] == 1) and not shift[128:].any()


def test_bf16_rounding_is_nearest_even(hp):
    rng = np.random.RandomState(1)
    vals = np.concatenate([rng.randn(2000).astype(np.float32), np.float32([0, -0.0, 1, 1.00390625, 1.01171875, 3.3895314e38])])
    for x in vals:
        b = hp.hp_to_bf16(float(x))
        u = int(np.float32(x).view(np.uint32))
        want = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF
        assert b == want
        back = hp.hp_from_bf16(b)
        assert np.float32(back).view(np.uint32) == (b << 16)


def test_split3_planes_reconstruct_the_weights(hp):
    """pack_split3 (the split-product path's weight layout: per 32-element K chunk three planes hi / mid / lo of [rows][32] bf16):
    hi + mid + lo reproduces every fp32 weight to within one unit of its 24th bit, hi is the weight's bf16 rounding, and the planes
    sit where the kernel's LDS-DMA addressing expects them."""
    rng = np.random.RandomState(9)
    Npad, K = 64, 96
    wp = (rng.randn(Npad, K) * np.exp(rng.uniform(-6, 3, (Npad, K)))).astype(np.float32)
    wp[0, :4] = [0.0, -0.0, 1.0, -3.0e-12]
    out = np.zeros((K // 32, 3, Npad, 32), np.uint16)                                  # chunk-major: [chunk][plane][row][32]
    hp.hp_pack_split3.argtypes = [f32p, C.c_int, C.c_int, C.POINTER(C.c_uint16)]
    hp.hp_pack_split3(_p(wp, f32p), Npad, K, _p(out, C.POINTER(C.c_uint16)))
    planes = (out.astype(np.uint32) << 16).view(np.float32).astype(np.float64)          # bf16 -> value
    total = planes.sum(axis=1).transpose(1, 0, 2).reshape(Npad, K)
    w64 = wp.astype(np.float64)
    assert np.all(np.abs(total - w64) <= np.abs(w64) * 2.0 ** -23 + 1e-300)
    u = wp.view(np.uint32).astype(np.uint64)
    hi_want = (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16).reshape(Npad, K // 32, 32)
    assert np.array_equal(out[:, 0, :, :].transpose(1, 0, 2), hi_want)


# ------------------------------------------------------------------------------------------ arena, tiles, stem
def test_arena_first_fit_never_overlaps_live_tensors(hp):
    rng = np.random.RandomState(11)
    for trial in range(30):
        n = int(rng.randint(1, 120))
        first = rng.randint(-1, 60, n).astype(np.int32)
        last = (first + rng.randint(0, 12, n)).astype(np.int32)
        need = (rng.randint(1, 5000, n) * 256).astype(np.uint64)
        off = np.zeros(n, np.uint64)
        total = hp.hp_arena(_p(first, i32p), _p(last, i32p), _p(need, u64p), n, _p(off, u64p))
        assert total == int((off + need).max())
        for a in range(n):
            for b in range(a + 1, n):
                live = not (last[a] < first[b] or last[b] < first[a])
                disjoint = off[a] + need[a] <= off[b] or off[b] + need[b] <= off[a]
                assert disjoint or not live, (trial, a, b)
        assert total <= int(need.sum())


def test_tile_choice_invariants_over_the_whole_network(hp):
    """choose_tile on every conv of the net (3 scales, both precisions): a legal shape, K groups divide the chunks per tap, a K split
    never exceeds the chunk count; the 23x23 long-K layers get the in-workgroup K groups; VNECT_FORCE_TILE / VNECT_PLAN parse strictly."""
    from vnect_amd.weights import CONV_LAYERS
    out = (C.c_int * 4)()
    seen = set()
    side = {"res2": 92, "res3": 46, "res4": 23, "res5": 23}     # output grid of each stage (368 / 4, / 8, / 16; res5 keeps 23)
    for bf16 in (0, 1):
        epr = 64 if bf16 else 32
        for name, k, cin, cout, _ in CONV_LAYERS:
            if k == 7:
                continue
            hw = 46 if name.startswith("res5c") else side[name[:4]]
            cs = -(-cin // epr) * epr
            cpt = cs // epr
            M = 3 * hw * hw
            hp.hp_choose_tile(M, cout, k * k, cpt, k * k * cs, 1, bf16, name.encode(), None, None, out)
            BM, BN, KG, ks = out
            assert (BM, BN, KG) in ((64, 64, 1), (64, 32, 2), (32, 32, 4)) and cpt % KG == 0 and 1 <= ks <= max(1, k * k * cpt // KG), name
            seen.add((BM, BN, KG, ks > 1))
    assert (64, 32, 2, False) in seen and (32, 32, 4, False) in seen and (64, 64, 1, True) in seen
    hp.hp_choose_tile(1587, 256, 9, 8, 2304, 1, 0, b"res4b_branch2b", b"64,64,1,5", None, out)
    assert list(out) == [64, 64, 1, 5]
    hp.hp_choose_tile(1587, 256, 9, 8, 2304, 1, 0, b"res4b_branch2b", b"48,64,1,5", None, out)   # not a shape the kernels have
    assert list(out) == [64, 32, 2, 1]
    hp.hp_choose_tile(1587, 256, 9, 8, 2304, 1, 0, b"res4b_branch2b", None, b"res4a_branch2b=64,64,1,2;res4b_branch2b=32,32,4,3", out)
    assert list(out) == [32, 32, 4, 3]
    hp.hp_choose_tile(1587, 256, 9, 8, 2304, 1, 0, b"branch2b", None, b"res4b_branch2b=32,32,4,3", out)   # a suffix is not a match
    assert list(out) == [64, 32, 2, 1]
    # the transposed conv (4 phases, N = 191, K = 4 taps x 256): 300 tiles of 64x64 at three scales -> the three-accumulator 64x96x2 shape
    # (200 tiles) where it is allowed, fp32 only, and only while it saves a round (one scale takes the 64x32x2 K groups as before, two scales fit one round of 64x64 tiles; four
    # scales need two rounds either way)
    dc = lambda fn, M, bf16=0, force=None, plan=None: (fn(M, 191, 4, 8 if not bf16 else 4, 1024, 4, bf16, b"res5c_deconv", force, plan, out), list(out))[1]
    assert dc(hp.hp_choose_tile96, 3 * 529) == [64, 96, 2, 1] and dc(hp.hp_choose_tile, 3 * 529) == [64, 64, 1, 1]
    assert dc(hp.hp_choose_tile96, 3 * 529, bf16=1) == [64, 64, 1, 1]
    assert dc(hp.hp_choose_tile96, 529) == [64, 32, 2, 1] and dc(hp.hp_choose_tile96, 2 * 529) == [64, 64, 1, 1] and dc(hp.hp_choose_tile96, 4 * 529) == [64, 64, 1, 1]
    assert dc(hp.hp_choose_tile96, 2 * 529, plan=b"res5c_deconv=64,96,2,1") == [64, 96, 2, 1]         # by plan, where the default would not pick it
    assert dc(hp.hp_choose_tile, 3 * 529, plan=b"res5c_deconv=64,96,2,1") == [64, 64, 1, 1]           # ... but never where the caller does not admit it (split-product handle, VNECT_NO_DECONV96)
    assert dc(hp.hp_choose_tile96, 3 * 529, plan=b"res5c_deconv=64,64,1,1") == [64, 64, 1, 1]
    hp.hp_choose_tile(1587, 256, 9, 8, 2304, 1, 0, b"res4b_branch2b", b"64,96,2,1", None, out)                 # ... and on no other
    assert list(out) == [64, 32, 2, 1]


def test_pair_head_split_only_where_it_saves_a_round(hp):
    """plan::pair_head_cols: the paired 1x1 launches of the net at 1 .. 6 scales.  Three scales: only res5a's pair (600 tiles of 64x64 = three
    rounds; 256 head channels as 200 tiles of 64x32x2 + 500 tiles = two and a half) -- and whatever it returns is a shape choose_tile
    really gives the 64x32x2 K groups in ONE round, fp32 only."""
    out = (C.c_int * 4)()
    pairs = {"res3a": (46, 128, 512, 256), "res4a": (23, 256, 1024, 512), "res5a": (23, 512, 1024, 1024), "res2a": (92, 64, 256, 64)}
    got = {}
    for S in range(1, 7):
        for name, (hw, ca, cb, K) in pairs.items():
            M = S * hw * hw
            assert hp.hp_pair_head_cols(M, ca, cb, K, 1) == 0
            c = hp.hp_pair_head_cols(M, ca, cb, K, 0)
            got[(S, name)] = c
            if c:
                assert c % 64 == 0 and 0 < c < ca
                hp.hp_choose_tile(M, c, 1, K // 32, K, 1, 0, b"head", None, None, out)
                assert list(out) == [64, 32, 2, 1] and 128 < -(-M // 64) * (c // 32) <= 256
                rounds = lambda t: -(-t // 256)
                mt, nt = -(-M // 64), (ca + cb) // 64
                assert rounds(mt * (nt - c // 64)) + 0.5 < rounds(mt * nt)
    assert got[(3, "res5a")] == 256
    assert [k for k, v in got.items() if v and k[0] <= 3] == [(3, "res5a")]


def test_stem_row_groups_and_frame_eligibility(hp):
    row0 = (C.c_uint8 * 93)()
    for S in range(1, 9):
        G = hp.hp_stem_groups(S, row0)
        r = list(row0[:G + 1])
        assert r[0] == 0 and r[-1] == 92 and all((4 if S >= 3 else 2) <= b - a <= 5 for a, b in zip(r, r[1:])), (S, r)
        if S <= 3:
            assert S * G * 4 <= 256     # one tile per CU, one round
    assert hp.hp_stem_groups(3, row0) == 21 and hp.hp_stem_groups(2, row0) == 32 and hp.hp_stem_groups(1, row0) == 46

    def fits(scales, bf16=0):
        s = np.array(scales + [1.0] * (8 - len(scales)), np.float64)
        return hp.hp_stem_frame_fits(_p(s, f64p), len(scales), 0, bf16)
    assert fits([1.0, 0.8, 0.6]) == 1 and fits([1, 0.85, 0.7]) == 1 and fits([1.0, 0.8, 0.6], 1) == 1 and fits([1.0, 0.5]) == 1
    assert fits([1.0, 0.4]) == 1        # (a tile only needs the part of its patch that lies inside the scaled image)
    assert fits([1.0, 0.3]) == 1        # two images: 2- and 3-row tiles, whose patches need fewer frame rows
    assert fits([1.0, 0.8, 0.3]) == 0   # three: 4- and 5-row tiles -- more frame rows than the kernel's LDS rectangle holds: the batch-tensor form runs instead
    assert fits([1.0, 0.15]) == 0


# ------------------------------------------------------------------------------------------ the same, sanitized
def test_hostplan_clean_under_asan_ubsan():
    if os.environ.get("VNECT_HOSTPLAN_SO"):
        pytest.skip("already inside the sanitized run")
    subprocess.check_call(["make", "-C", CSRC, "hostplan_asan"], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "vnect_amd", "lib", "libvnect_hostplan_asan.so")
    libasan = subprocess.check_output(["g++", "-print-file-name=libasan.so"], text=True).strip()
    libubsan = subprocess.check_output(["g++", "-print-file-name=libubsan.so"], text=True).strip()
    env = dict(os.environ, VNECT_HOSTPLAN_SO=so, LD_PRELOAD=libasan + ":" + libubsan, VNECT_ORACLE_THREADS="8",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.abspath(__file__)],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail


def test_tile_enumerator_runs_on_the_committed_layer_table():
    """tools/tile_enum.py (the quantisation argument) on the newest committed layer table: it parses every conv launch,
    no plan beats the ideal (the launch's blocks over 1 024 SIMDs), the 16-granular search contains the 32-granular one, and the plan in
    use is never better than the best the enumerator finds by more than the per-tile cost it charges."""
    import glob
    import re
    import subprocess
    import sys
    root = ROOT
    tabs = sorted(glob.glob(os.path.join(root, "profiles", "r0[0-9]_layer_table.txt")))
    assert tabs, "no committed layer table"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "tile_enum.py"), tabs[-1]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [ln for ln in r.stdout.splitlines() if re.match(r"res\d", ln)]
    assert len(rows) >= 36, len(rows)                      # every conv launch of a three-scale frame behind the stem
    for ln in rows:
        f = [x.strip() for x in ln.split("|")]
        ideal, inuse = (float(x) for x in f[1].split())
        b32, b16 = float(f[2].split()[-1]), float(f[3].split()[-1])
        assert ideal <= b16 + 1e-9 and b16 <= b32 + 1e-9, ln   # nothing beats the ideal; the finer search contains the coarser one
        assert ideal <= inuse + 1e-9, ln
    assert "upper bound" in r.stdout.splitlines()[-1]
