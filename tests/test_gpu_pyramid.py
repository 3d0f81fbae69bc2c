"""GPU (one MI355X), through the C ABI: pyramid sharding (BASELINE.json configs[3]) as far as one device can run it --
rank handles reassembled on the host, the peer-write exchange between processes, a 1-rank RCCL communicator.  The multi-device
tests are tests/test_gpu_multigpu.py."""
import json
import os

import numpy as np
import pytest

from tests.gpu_common import BASELINE_SCALES, G, OUT, T0, _EndToEnd, _handle, _log, _native, _round_bf16  # noqa: F401

pytestmark = pytest.mark.gpu


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
