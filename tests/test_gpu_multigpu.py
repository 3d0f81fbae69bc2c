"""GPU, MORE THAN ONE MI355X: parity of the two N > 1 paths (BASELINE.json configs[3] / configs[4]), armed for the first
multi-GPU box that runs `pytest -m gpu`.  On a one-GPU box the `needs_N_gpus` tests SKIP (the reason says so) and the `rehearsal`
tests run the SAME comparison code with every rank on device 0 over gloo, so that the day the devices exist the only new thing is
the hardware: RCCL with more than one rank, xGMI, cross-device visibility of the peer-write exchange.

The reference's seam: the S images of a frame's batch meet only in the merge (/root/reference/src/estimator.py:100-129), and one
estimator per video lives in a process of its own (/root/reference/run_estimator_ps.py:120-129).

What each armed test proves for the first time ("still unproven"):
* test_needs_3_gpus_pyramid_both_exchange_forms_bit_equal -- RCCL bootstrap + ncclAllGather with 3 ranks; hipIpcOpenMemHandle of a
  PEER device's fine-grained block; cross-device visibility of the exchange's write-through stores and flags; both forms bit-equal to
  each other and to three rank handles on device 0 whose maps are stacked on the host;
* test_needs_2_gpus_p2p_missing_peer_fails_the_frame -- the bounded wait across devices: VNECT_E_COMM, never a hang;
* test_needs_2_gpus_stream_replicas_rccl -- torch's "nccl" process group with 2 ranks (barrier + max-reduce), per-GPU rate against the
  N = 1 line, every stream's joints equal to a single-GPU run of the same seeds.
"""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCALES = [1.0, 0.8, 0.6]


def _ndev():
    import torch
    return torch.cuda.device_count()   # (counting devices does not initialise the GPU on this image)


def needs(n):
    return pytest.mark.skipif(_ndev() < n, reason="needs %d HIP devices, this box has %d: SKIPPED here, armed for a multi-GPU box" % (n, _ndev()))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT",
                                                             "VNECT_BENCH_BACKEND", "VNECT_BENCH_DEVICE", "VNECT_BENCH_WORKER")}
    env.update(extra)
    return env


def _bench(argv, env, timeout=1500):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    return json.loads(lines[0])


def _dump(prefix, leg, rank):
    d = np.load("%s.%s.rank%d.npz" % (prefix, leg, rank))
    return d["j2"], d["j3"], int(d["device"]), int(d["stream"])


def _single_gpu_stream(weights, stream):
    """What bench.py's --dump-joints frames must be: the same seeds through a plain 3-scale handle on device 0."""
    import bench
    from tests import helpers
    from vnect_amd import _native
    from vnect_amd.parallel import stream_seed
    h = _native.Handle(SCALES, device=0, num_frame_slots=8)
    h.set_weights(weights)
    h.finalize()
    for k in range(8):
        h.upload_frame(k, helpers.synth_frame(stream_seed(stream, k)))
    js = [h.infer_resident(k, bench.DUMP_T0 + k / 30, bench.DUMP_T0 + k / 30 + 1e-3) for k in range(8)]
    h.close()
    return np.stack([a for a, _ in js]), np.stack([b for _, b in js])


def _host_stacked_pyramid(weights):
    """The sharded job's reference: three rank handles (one scale each: the launch plan a pyramid rank runs) on device 0, no exchange --
    pre-processing + vnect_forward, maps stacked on the host in rank order, one handle's post-processing over the stack with its filter
    chain in lockstep.  (A plain 3-scale handle is NOT the bit-level reference: its 3-image launches may split K differently, so its
    maps agree to fp32 rounding only -- tests/test_gpu_pyramid.py::test_pyramid_shards_reassemble holds that to 1e-5.)"""
    import bench
    from tests import helpers
    from vnect_amd import _native
    from vnect_amd.parallel import stream_seed
    ranks = [_native.Handle(SCALES, device=0, pyramid=(r, 3)) for r in range(3)]
    for h in ranks:
        h.set_weights(weights)
        h.finalize()
    j2s, j3s = [], []
    for k in range(8):
        frame = helpers.synth_frame(stream_seed(0, k))
        maps = []
        for h in ranks:
            b, scaler, (ox, oy) = h.preprocess(frame)
            maps.append(h.forward(b)[0])
        t = bench.DUMP_T0 + k / 30
        j2, j3 = ranks[0].postprocess(np.stack(maps), t, t + 1e-3, scaler, ox, oy)
        j2s.append(j2), j3s.append(j3)
    for h in ranks:
        h.close()
    return np.stack(j2s), np.stack(j3s)


def _check_self_explaining(d, n, same_device, replicas):
    """Round 6: an N > 1 line must explain itself with nobody there to debug it -- each rank's own rate / latency / conv-stack time /
    shader clock under load, where its host thread sat, and the N = 1 loop of the same job."""
    pr = d["per_rank"]
    assert [r["rank"] for r in pr] == list(range(n)) and len(d["ranks"]) == n
    for r in pr:
        assert r["frames_per_s"] > 0 and r["own_elapsed_s"] > 0 and r["latency_ms"]["p50"] > 0 and r["latency_ms"]["p95"] >= r["latency_ms"]["p50"]
        assert abs(r["frames_per_s"] - d["steps"] / r["own_elapsed_s"]) <= 1e-2 * r["frames_per_s"] and r["closing_barrier_us"] >= 0
        assert 0 < r["conv_stack_ms"] <= r["frame_ms_hip_events"] * 1.05
        assert 800 < r["shader_clock_mhz"] < 2600, r        # MI355X: up to 2.4 GHz; the conv stack holds ~2.1 under load
        assert "bound" in r["host_binding"] and ("affinity" in r["host_binding"] or "reason" in r["host_binding"])
    # `value` is N K / the SLOWEST rank's time (barrier to barrier): no rank's own rate is below value / N by more than the barrier's cost
    if replicas:
        assert min(r["frames_per_s"] for r in pr) >= 0.97 * d["value"] / n, (d["value"], [r["frames_per_s"] for r in pr])
    for r in d["ranks"]:
        hb = r["host_binding"]
        assert isinstance(hb["bound"], bool) and (hb["bound"] is False or hb["n_cpus"] >= 1)
        if hb.get("bdf") is not None and "bdf_is_the_hip_device" in hb:
            assert hb["bdf_is_the_hip_device"] is True, hb   # the sysfs chain and HIP name the same device
    n1 = d["n1_same_job"]
    assert n1 and n1["rank"] == 0 and n1["value"] > 0 and n1["steps"] == d["steps"] and n1["latency_ms"]["p50"] > 0
    assert 0 < n1["conv_stack_ms"] <= n1["frame_ms_hip_events"] * 1.05 and 800 < n1["shader_clock_mhz"] < 2600, n1
    # ... and tools/explain_scale.py can read the line (the attribution an unattended run gets)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import explain_scale
    text = explain_scale.explain(d)
    assert "attribution:" in text and ("same-job N = 1 loop" in text), text
    # (whether a rank COULD bind depends on what sysfs the container shows: a rank that could not says why -- never a parity failure)
    assert all(r["host_binding"]["bound"] or r["host_binding"].get("reason") for r in pr), [r["host_binding"] for r in pr]
    assert "probes_off=1" in d["build"] and "test_hooks=0" in d["build"]


def _check_replicas(weights, d, prefix, n, same_device):
    assert d["n_gpus"] == n and d["backend_ranks"] == n and d["scaling"] == "weak" and len(d["ranks"]) == n
    _check_self_explaining(d, n, same_device, replicas=True)
    if not same_device:
        assert d["rccl_ranks"] == n and d["backend"] == "nccl"
        assert len({r["pci_bus_id"] for r in d["ranks"]}) == n and sorted(r["device"] for r in d["ranks"]) == list(range(n)), d["ranks"]
    for r in range(n):
        j2, j3, dev, stream = _dump(prefix, "replica", r)
        assert stream == r and dev == (0 if same_device else r)
        r2, r3 = _single_gpu_stream(weights, r)
        assert np.array_equal(j2, r2) and np.array_equal(j3, r3), "rank %d: its stream differs from a single-GPU run of the same seeds" % r
    a, b = _dump(prefix, "replica", 0), _dump(prefix, "replica", 1)
    assert not np.array_equal(a[0], b[0])   # two different videos (seeds 1234 + 1000 * rank): the ranks did not all run stream 0


def _check_pyramid(weights, prefix, legs):
    ref2, ref3 = _host_stacked_pyramid(weights)
    for leg in legs:
        outs = [_dump(prefix, leg, r) for r in range(3)]
        for r in range(1, 3):   # filters run redundantly on every rank: every rank returns the joints
            assert np.array_equal(outs[0][0], outs[r][0]) and np.array_equal(outs[0][1], outs[r][1]), (leg, r)
        assert np.array_equal(outs[0][0], ref2) and np.array_equal(outs[0][1], ref3), \
            "%s: the sharded job's joints differ from three rank handles on device 0 with host-stacked maps" % leg


# ------------------------------------------------------------------------------------------ armed: need the devices
@needs(3)
def test_needs_3_gpus_pyramid_both_exchange_forms_bit_equal(weights, tmp_path):
    """configs[3] on hardware: `bench.py --gpus 3 --pyramid-both` -- 3 RCCL ranks on 3 distinct devices, the all-gather form and the
    peer-write form in one job, each bit-equal on every rank to the host-stacked reference on device 0 (hence to each other)."""
    prefix = str(tmp_path / "j")
    d = _bench(["--gpus", "3", "--pyramid-both", "--steps", "30", "--warmup", "5", "--cpu-seconds", "0", "--dump-joints", prefix], _clean_env())
    assert d["rccl_ranks"] == 3 and d["backend"] == "nccl" and d["n_gpus"] == 3 and d["scaling"] == "strong"
    assert len({r["pci_bus_id"] for r in d["ranks"]}) == 3 and sorted(r["device"] for r in d["ranks"]) == [0, 1, 2], d["ranks"]
    assert d["exchange"].startswith("rccl") and d["pyramid_p2p"] and d["pyramid_p2p"]["value"] > 0
    _check_self_explaining(d, 3, same_device=False, replicas=False)
    assert [r["rank"] for r in d["pyramid_p2p"]["per_rank"]] == [0, 1, 2]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "multigpu_pyramid_both.json"), "w") as f:
        json.dump(d, f, indent=1)
    _check_pyramid(weights, prefix, ["pyramid_rccl", "pyramid_p2p"])


P2P_MISSING_PEER = r"""
import os, sys, time
sys.path.insert(0, %r)
from vnect_amd import _native
from vnect_amd.weights import synthetic_weights
from tests import helpers
rank, dev, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
def note(msg):
    sys.stderr.write("[missing-peer rank %%d] %%s\n" %% (rank, msg)); sys.stderr.flush()
h = _native.Handle([1.0, 0.7], device=dev, pyramid=(rank, 2), exchange=_native.XCHG_P2P)
h.set_weights(synthetic_weights())
h.finalize()
open(os.path.join(d, "blob%%d.tmp" %% rank), "wb").write(h.p2p_export())
os.rename(os.path.join(d, "blob%%d.tmp" %% rank), os.path.join(d, "blob%%d" %% rank))
t0 = time.time()
while not all(os.path.exists(os.path.join(d, "blob%%d" %% r)) for r in range(2)):
    assert time.time() - t0 < 300
    time.sleep(0.05)
note("blobs there")
h.p2p_init(rank, 2, [open(os.path.join(d, "blob%%d" %% r), "rb").read() for r in range(2)])
note("p2p_init done")
open(os.path.join(d, "ready%%d" %% rank), "w").write("x")
t0 = time.time()
while not all(os.path.exists(os.path.join(d, "ready%%d" %% r)) for r in range(2)):   # nobody closes a block a peer is still mapping
    assert time.time() - t0 < 300
    time.sleep(0.05)
if rank == 0:
    try:
        h.infer(helpers.synth_frame(3), 1.7e9, 1.7e9)     # rank 1 never submits this frame
        print("RESULT no-error", flush=True)
    except _native.VnectError as e:
        print("RESULT code %%d" %% e.code, flush=True)
    note("frame returned")
    open(os.path.join(d, "done"), "w").write("x")
else:
    t0 = time.time()
    while not os.path.exists(os.path.join(d, "done")):     # stays alive (its block stays mapped) and never exchanges
        assert time.time() - t0 < 600
        time.sleep(0.05)
    print("RESULT idle", flush=True)
note("closing")
h.close()
note("closed")
"""


def _missing_peer(tmp_path, devices):
    script = tmp_path / "p2p_missing_peer.py"
    script.write_text(P2P_MISSING_PEER % ROOT)
    env = _clean_env(VNECT_XCHG_SPINS="20000")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    logs = [open(os.path.join(ROOT, "gpurun_out", "missing_peer_rank%d.log" % r), "w") for r in range(2)]   # a hang leaves its trace here
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(devices[r]), str(tmp_path)], env=env, stdout=subprocess.PIPE,
                              stderr=logs[r], text=True) for r in range(2)]
    t0 = time.time()
    res = []
    try:
        for r, p in enumerate(procs):
            o, _ = p.communicate(timeout=150)
            assert p.returncode == 0, open(logs[r].name).read()[-3000:]
            res.append([ln for ln in o.splitlines() if ln.startswith("RESULT")][-1])
    finally:
        for p in procs:          # the exact processes this test started
            if p.poll() is None:
                p.kill()
        for f in logs:
            f.close()
    from vnect_amd import _native
    assert res[0] == "RESULT code %d" % _native.E_COMM and res[1] == "RESULT idle", res
    return time.time() - t0


@needs(2)
def test_needs_2_gpus_p2p_missing_peer_fails_the_frame(tmp_path):
    """The peer-write exchange's bounded wait ACROSS devices: rank 0 (device 0) submits a frame, rank 1 (device 1, block IPC-mapped,
    alive) never does -- VNECT_E_COMM after the bound, no hang, both processes exit cleanly."""
    _missing_peer(tmp_path, [0, 1])


@needs(2)
def test_needs_2_gpus_stream_replicas_rccl(weights, tmp_path):
    """configs[4] with two ranks: `bench.py --gpus 2` (self-spawned, backend nccl): rccl_ranks == 2 on two distinct devices, each rank's
    stream bit-equal to a single-GPU run of the same seeds, and the per-GPU rate within 10 % of the N = 1 line of the same call."""
    prefix = str(tmp_path / "j")
    common = ["--steps", "200", "--warmup", "20", "--cpu-seconds", "0", "--no-aux"]
    one = _bench(["--gpus", "1"] + common, _clean_env())
    two = _bench(["--gpus", "2", "--dump-joints", prefix] + common, _clean_env())
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "multigpu_replicas_2.json"), "w") as f:
        json.dump({"n1": one, "n2": two}, f, indent=1)
    _check_replicas(weights, two, prefix, 2, same_device=False)
    per_gpu = two["value"] / 2
    assert abs(per_gpu - one["value"]) <= 0.10 * one["value"], (per_gpu, one["value"])
    assert abs(per_gpu - two["n1_same_job"]["value"]) <= 0.10 * two["n1_same_job"]["value"], (per_gpu, two["n1_same_job"])


# ------------------------------------------------------------------------------------------ rehearsals: the same checks on ONE device
def test_rehearsal_one_device_two_replicas_gloo(weights, tmp_path):
    """The comparison code of test_needs_2_gpus_stream_replicas_rccl with both ranks on device 0 over gloo (RCCL refuses two ranks on one
    device): spawner, --dump-joints, every stream against a single-GPU run of its seeds.  No rate is asserted."""
    prefix = str(tmp_path / "j")
    d = _bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--cpu-seconds", "0", "--no-aux", "--dump-joints", prefix],
               _clean_env(VNECT_BENCH_BACKEND="gloo", VNECT_BENCH_DEVICE="0"))
    assert d["backend"] == "gloo" and d["rccl_ranks"] is None
    _check_replicas(weights, d, prefix, 2, same_device=True)


def test_rehearsal_one_device_pyramid_p2p_gloo(weights, tmp_path):
    """The comparison code of test_needs_3_gpus_pyramid_both_exchange_forms_bit_equal with the three ranks on device 0 (three processes,
    IPC-mapped blocks, gloo): the peer-write leg only -- the line says why -- against the host-stacked reference."""
    prefix = str(tmp_path / "j")
    d = _bench(["--gpus", "3", "--pyramid-both", "--steps", "20", "--warmup", "5", "--cpu-seconds", "0", "--dump-joints", prefix],
               _clean_env(VNECT_BENCH_BACKEND="gloo", VNECT_BENCH_DEVICE="0"))
    assert d["backend_ranks"] == 3 and d["exchange"].startswith("p2p") and "RCCL leg" in d["note"]
    _check_self_explaining(d, 3, same_device=True, replicas=False)
    _check_pyramid(weights, prefix, ["pyramid_p2p"])


def test_rehearsal_one_device_p2p_missing_peer_two_processes(tmp_path):
    """The two-process form of the missing-peer test with both ranks on device 0 (the one-process form is in tests/test_gpu_pyramid.py)."""
    _missing_peer(tmp_path, [0, 0])
