"""CPU: the N>1 host logic of bench.py (stream sharding, barrier, max-over-ranks) with world_size 2 over gloo."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, time, json
sys.path.insert(0, %r)
from vnect_amd.parallel import Group, aggregate_rate, stream_seed
from tests import helpers
g = Group("gloo")
assert g.world == 2
# each rank owns an independent stream: different seeds -> different frames, no data exchange
f = helpers.synth_frame(stream_seed(g.rank, 0), 16, 16)
g.barrier()
elapsed = 0.5 + 0.25 * g.rank          # rank 1 is the slow one
m = g.max_over_ranks(elapsed)
assert m == 0.75, m
rate = aggregate_rate(g.world, 30, m)
print(json.dumps({"rank": g.rank, "max": m, "rate": rate, "sum": int(f.sum())}), flush=True)
g.close()
"""


def test_two_rank_stream_replicas_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29517")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=240)
        assert p.returncode == 0, e[-2000:]
        outs.append(eval(o.strip().splitlines()[-1].replace("true", "True")))
    assert {o["rank"] for o in outs} == {0, 1}
    assert all(o["max"] == 0.75 and abs(o["rate"] - 2 * 30 / 0.75) < 1e-9 for o in outs)
    assert outs[0]["sum"] != outs[1]["sum"]   # the two streams are different videos


def test_world_one_needs_no_process_group():
    from vnect_amd.parallel import Group, aggregate_rate, stream_seed
    env_keys = ("RANK", "WORLD_SIZE", "LOCAL_RANK")
    saved = {k: os.environ.pop(k, None) for k in env_keys}
    try:
        g = Group("gloo")
        assert g.world == 1 and g.max_over_ranks(1.5) == 1.5
        g.barrier()
        g.close()
        assert stream_seed(3, 7) == 1234 + 3000 + 7
        assert aggregate_rate(8, 100, 2.0) == 400.0
    finally:
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v


PYRAMID_WORKER = r"""
import os, sys, json
sys.path.insert(0, %r)
from vnect_amd.parallel import Group, PyramidJob, stream_seed
from tests import helpers

class StubHandle:
    # the pyramid part of _native.Handle's surface, recording what the host logic does with it
    log = []
    def __init__(self, rank, world, exchange):
        self.rank, self.world, self.exchange = rank, world, exchange
        self.frames, self.inferred, self.connected = {}, 0, None
    @staticmethod
    def comm_unique_id():
        StubHandle.log.append("uid")
        return bytes(range(128))
    def comm_init(self, rank, world, uid):
        assert (rank, world) == (self.rank, self.world) and uid == bytes(range(128))
        self.connected = "rccl"
    def p2p_export(self):
        return bytes([self.rank]) * 128
    def p2p_init(self, rank, world, blobs):
        assert (rank, world) == (self.rank, self.world) and len(blobs) == world
        assert [b[0] for b in blobs] == list(range(world)) and all(len(b) == 128 for b in blobs)
        self.connected = "p2p"
    def upload_frame(self, slot, frame):
        self.frames[slot] = int(frame.sum())
    def infer_resident(self, slot, t2d, t3d):
        assert self.connected and slot in self.frames and t3d > t2d
        self.inferred += 1
        return ("joints", slot)

g = Group("gloo")
assert g.world == 3
scales = [1.0, 0.8, 0.6]
res = {}
for exchange in ("rccl", "p2p"):
    StubHandle.log = []
    job = PyramidJob(g, scales, StubHandle, exchange)
    assert job.scale == scales[g.rank] and job.handle.connected == exchange
    # only rank 0 makes the communicator id; everybody receives it
    assert (StubHandle.log == ["uid"]) == (g.rank == 0 and exchange == "rccl")
    # every rank uploads the SAME stream 0
    job.upload([helpers.synth_frame(stream_seed(0, k), 16, 16) for k in range(4)])
    g.barrier()
    (last, t) = job.run(10, 4, 100.0)          # the timed loop ...
    (last, t) = job.run(5, 4, t)               # ... and the profiling loop: EVERY rank takes part in both
    assert last == ("joints", 0) and job.handle.inferred == 15 and abs(t - (100.0 + 15 / 30)) < 1e-9
    res[exchange] = [job.handle.frames[k] for k in range(4)]
    m = g.max_over_ranks(1.0 + g.rank)
    assert m == 3.0
try:
    PyramidJob(g, [1.0, 0.7], StubHandle)
    raise SystemExit("a 2-scale pyramid on 3 ranks must be refused")
except ValueError:
    pass
print(json.dumps({"rank": g.rank, "frames": res}), flush=True)
g.close()
"""


def test_three_rank_pyramid_host_logic_gloo(tmp_path):
    """bench.py --pyramid's host side (vnect_amd.parallel.PyramidJob) with world_size 3 over gloo and a stub handle: rank -> scale
    mapping, the ncclUniqueId made by rank 0 and broadcast, the all-gather of the p2p export blobs, the same stream on every
    rank, and every rank taking part in the timed AND the profiling loop (each inference contains the exchange)."""
    script = tmp_path / "pyramid_worker.py"
    script.write_text(PYRAMID_WORKER % ROOT)
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29519")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=240)
        assert p.returncode == 0, e[-2000:]
        outs.append(eval(o.strip().splitlines()[-1]))
    assert {o["rank"] for o in outs} == {0, 1, 2}
    assert outs[0]["frames"] == outs[1]["frames"] == outs[2]["frames"]   # one stream, seen by all ranks


SPAWN_STUB = r"""
import os, sys, json
sys.path.insert(0, %r)
from vnect_amd.parallel import Group
assert os.environ["VNECT_BENCH_SPAWNED"] == "1" and os.environ["MASTER_ADDR"] == "127.0.0.1"
g = Group("gloo")                      # the ranks the parent started rendezvous by themselves
n = g.count_ranks()                    # a real all-reduce over the backend
seen = g.all_gather_object({"rank": g.rank, "device": g.local_rank})
if g.rank == 0:
    print("rank 0 chatter that is not the result line", flush=True)
    print(json.dumps({"n_gpus": g.world, "backend_ranks": n, "ranks": seen, "argv": sys.argv[1:]}), flush=True)
else:
    print("rank %%d must not reach the parent's stdout" %% g.rank, flush=True)
g.close()
sys.exit(int(os.environ.get("STUB_FAIL_RANK", "-1")) == g.rank and 7 or 0)
"""


def _run_bench_bare(tmp_path, extra_env, gpus=2):
    stub = tmp_path / "stub_worker.py"
    stub.write_text(SPAWN_STUB % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(VNECT_BENCH_WORKER=str(stub), **extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1"],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher (the driver's command shape): the parent starts 2 rank processes with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's JSON line -- and only that line -- and exits 0."""
    import json
    r = _run_bench_bare(tmp_path, {})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["backend_ranks"] == 2
    assert [(x["rank"], x["device"]) for x in out["ranks"]] == [(0, 0), (1, 1)]
    assert out["argv"] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]   # the children get the parent's arguments
    assert "rank 1 must not reach" in r.stderr and "rank 0 chatter" in r.stderr


def test_bench_spawner_reports_the_worst_return_code(tmp_path):
    r = _run_bench_bare(tmp_path, {"STUB_FAIL_RANK": "1"})
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])


def test_bench_more_ranks_than_devices_fails_at_once():
    """`python bench.py --gpus 2` (no launcher, backend "nccl") on a machine with fewer than 2 HIP devices -- this container has none,
    a one-GPU box has one: non-zero exit at once, the message names the missing device, no rank is started, nothing on stdout.
    Under a launcher's environment the rank itself refuses.  (The GPU-box twin is tests/test_gpu_multirank.py.)"""
    import time
    try:
        import torch
        have = torch.cuda.device_count()
    except Exception:
        have = 0
    if have >= 2:
        import pytest
        pytest.skip("two devices here")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "VNECT_BENCH_BACKEND",
                                                            "VNECT_BENCH_DEVICE", "VNECT_BENCH_WORKER")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert r.returncode != 0 and not r.stdout.strip(), (r.returncode, r.stdout)
    assert "HIP device %d is missing" % have in r.stderr and "exposes %d" % have in r.stderr, r.stderr[-1500:]
    assert time.time() - t0 < 300   # (a torch import in a throw-away child: seconds, minutes only in a container that has never imported torch)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=dict(env, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert r.returncode != 0 and "needs HIP device 1" in r.stderr and not r.stdout.strip(), (r.returncode, r.stderr[-1500:])


def test_one_rank_process_group_on_request():
    """Group(always_init=True) builds the process group for a world of ONE too (what a one-rank pyramid rehearsal needs: the same
    all-reduce / broadcast code path as the 3-rank job), and a one-scale PyramidJob connects through it."""
    code = r'''
import os, sys, json
sys.path.insert(0, %r)
from vnect_amd.parallel import Group, PyramidJob
g = Group("gloo", always_init=True)
assert g.world == 1 and g._dist is not None and g.count_ranks() == 1 and g.max_over_ranks(2.5) == 2.5
calls = []
class H:
    @staticmethod
    def comm_unique_id(): return b"u" * 128
    def comm_init(self, r, w, uid): calls.append(("comm_init", r, w, uid))
    def upload_frame(self, k, f): calls.append(("upload", k))
    def infer_resident(self, slot, a, b): calls.append(("infer", slot)); return slot, None
job = PyramidJob(g, [1.0], lambda r, w, ex: H(), "rccl")
assert calls == [("comm_init", 0, 1, b"u" * 128)] and job.scale == 1.0
job.upload(["f0", "f1"]); out, t = job.run(3, 2, 10.0)
assert [c for c in calls if c[0] == "infer"] == [("infer", 0), ("infer", 1), ("infer", 0)]
g.close()
print("ok")
''' % ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
