"""Shared pieces of the GPU parity tests (tests/test_gpu_*.py): handle builders, the log helper and the end-to-end gate.

Tolerances: conv stack fp32 vs oracle fp32: max|d| <= 1e-4 * max|map| (measured ~1e-6);
pre-processing and post-processing on identical inputs: bit-exact; joints_2d equal or within the tie
rule; joints_3d |d| <= 0.05 mm + 1e-4*|v|.
"""
import json
import os

import numpy as np

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


def _log(name, obj):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, indent=1)


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


def _round_bf16(a):
    """float32 -> nearest-even bfloat16 -> float32 (what the bf16 path's per-layer rounding does to a value)."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(a))
