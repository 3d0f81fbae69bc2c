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
# the host-side hand-off bench.py's same-job N = 1 reference uses: rank 0 works alone, rank 1 waits on the STORE (no collective enqueued)
t0 = time.time()
if g.rank == 0:
    time.sleep(0.6)
how = g.host_handoff("n1_done", release=g.rank == 0)
waited = time.time() - t0
g.barrier()
print(json.dumps({"rank": g.rank, "max": m, "rate": rate, "sum": int(f.sum()), "how": how, "waited": waited}), flush=True)
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
    by = {o["rank"]: o for o in outs}
    assert by[0]["how"] == by[1]["how"] == "store" and by[1]["waited"] >= 0.5   # rank 1 sat in the store wait while rank 0 "ran alone"


def test_world_one_needs_no_process_group():
    from vnect_amd.parallel import Group, aggregate_rate, stream_seed
    env_keys = ("RANK", "WORLD_SIZE", "LOCAL_RANK")
    saved = {k: os.environ.pop(k, None) for k in env_keys}
    try:
        g = Group("gloo")
        assert g.world == 1 and g.max_over_ranks(1.5) == 1.5 and g.host_handoff("x", release=True) == "none"
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


# ---------------------------------------------------------------------------------------------- host placement of a rank (round 6)
def _fake_sysfs(root, gpus, cpu_nodes=2):
    """A sysfs tree with `cpu_nodes` CPU nodes in front of the GPU nodes: gpus = [(domain, bus, dev, fn, local_cpulist, numa_node), ...]."""
    base = os.path.join(root, "class", "kfd", "kfd", "topology", "nodes")
    n = 0
    for _ in range(cpu_nodes):
        os.makedirs(os.path.join(base, str(n)))
        open(os.path.join(base, str(n), "properties"), "w").write("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
        n += 1
    for dom, bus, dev, fn, cpus, numa in gpus:
        os.makedirs(os.path.join(base, str(n)))
        open(os.path.join(base, str(n), "properties"), "w").write(
            "cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain %d\n" % ((bus << 8) | (dev << 3) | fn, dom))
        d = os.path.join(root, "bus", "pci", "devices", "%04x:%02x:%02x.%x" % (dom, bus, dev, fn))
        os.makedirs(d)
        if cpus is not None:
            open(os.path.join(d, "local_cpulist"), "w").write(cpus + "\n")
            open(os.path.join(d, "numa_node"), "w").write("%d\n" % numa)
        n += 1


def test_rank_binding_parser_on_a_fake_sysfs(tmp_path):
    """vnect_amd.parallel: where a rank's host thread goes, decided from sysfs alone (no GPU call): cpulist parsing, the KFD nodes ->
    PCI address -> local_cpulist chain, *_VISIBLE_DEVICES reordering, the intersection with the process's own affinity."""
    from vnect_amd import parallel as P
    assert P.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and P.parse_cpulist("") == [] and P.parse_cpulist(" 5 ") == [5]
    assert P.format_cpulist([11, 0, 1, 2, 3, 8, 10]) == "0-3,8,10-11" and P.format_cpulist([]) == ""
    for text in ("0-63,128-191", "7", "0,2,4,6"):
        assert P.format_cpulist(P.parse_cpulist(text)) == text
    root = str(tmp_path / "sys")
    gpus = [(0, 0x05, 0, 0, "0-63,128-191", 0), (0, 0x15, 0, 0, "0-63,128-191", 0), (0, 0x85, 0, 0, "64-127,192-255", 1),
            (1, 0x95, 1, 2, "64-127,192-255", 1), (0, 0xa5, 0, 0, None, -1)]
    _fake_sysfs(root, gpus)
    assert P.gpu_bdfs(root) == ["0000:05:00.0", "0000:15:00.0", "0000:85:00.0", "0001:95:01.2", "0000:a5:00.0"]   # CPU nodes skipped
    b = P.rank_binding(2, sysfs=root, env={})
    assert b["bdf"] == "0000:85:00.0" and b["numa_node"] == 1 and b["cpus"] == list(range(64, 128)) + list(range(192, 256))
    # the cgroup / taskset limit is respected: only local cores the process may already use
    b = P.rank_binding(3, sysfs=root, env={}, allowed=range(100, 200))
    assert b["bdf"] == "0001:95:01.2" and b["cpus"] == list(range(100, 128)) + list(range(192, 200))
    b = P.rank_binding(0, sysfs=root, env={}, allowed=[70, 71])
    assert b["cpus"] is None and "affinity" in b["reason"] and b["local_cpulist"] == "0-63,128-191"
    # *_VISIBLE_DEVICES: ROCr filters first, HIP indexes into what is left
    assert P.visible_order({}) is None and P.visible_order({"HIP_VISIBLE_DEVICES": "3,1"}) == [3, 1]
    assert P.visible_order({"ROCR_VISIBLE_DEVICES": "4,2,0", "HIP_VISIBLE_DEVICES": "1,0"}) == [2, 4]
    assert P.visible_order({"HIP_VISIBLE_DEVICES": "GPU-abcdef"}) == "unknown"
    assert P.rank_binding(0, sysfs=root, env={"HIP_VISIBLE_DEVICES": "2,3"})["bdf"] == "0000:85:00.0"
    assert P.rank_binding(0, sysfs=root, env={"HIP_VISIBLE_DEVICES": "GPU-abcdef"})["cpus"] is None
    # failures say why and never raise: no local_cpulist, a device sysfs does not show, a machine without KFD
    assert "local_cpulist" in P.rank_binding(4, sysfs=root, env={})["reason"]
    assert "not among" in P.rank_binding(7, sysfs=root, env={})["reason"]
    assert "no KFD" in P.rank_binding(0, sysfs=str(tmp_path / "nothing"), env={})["reason"]


def test_bind_rank_applies_and_reports(tmp_path):
    """bind_rank on a fake sysfs whose device is local to a SUBSET of this process's cores: the affinity mask is narrowed (in a child
    process), the report says so; --no-bind and a machine without GPUs leave it alone."""
    avail = sorted(os.sched_getaffinity(0))
    if len(avail) < 2:
        import pytest
        pytest.skip("one core")
    from vnect_amd import parallel as P
    root = str(tmp_path / "sys")
    half = avail[: len(avail) // 2]
    _fake_sysfs(root, [(0, 0x05, 0, 0, P.format_cpulist(half), 0)])
    code = r"""
import os, sys, json
sys.path.insert(0, %r)
from vnect_amd import parallel as P
rep = P.bind_rank(0, enable=True, sysfs=%r)
print(json.dumps({"rep": rep, "now": sorted(os.sched_getaffinity(0))}))
""" % (ROOT, root)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["now"] == half and out["rep"]["bound"] is True and out["rep"]["n_cpus"] == len(half) and out["rep"]["narrowed_from"] == len(avail)
    assert out["rep"]["bdf"] == "0000:05:00.0" and out["rep"]["affinity"] == P.format_cpulist(half)
    rep = P.bind_rank(0, enable=False, sysfs=root)
    assert rep["bound"] is False and rep["reason"] == "--no-bind" and sorted(os.sched_getaffinity(0)) == avail
    rep = P.bind_rank(0, enable=True, sysfs=str(tmp_path / "nothing"))
    assert rep["bound"] is False and "no KFD" in rep["reason"] and sorted(os.sched_getaffinity(0)) == avail


def test_rebind_when_hip_orders_devices_differently(tmp_path):
    """bind_rank trusts the KFD node order; once HIP is up the rank knows its device's PCI bus.  If the two disagree, the calling thread is moved
    to the right device's cores and the report says so (in a child process: the mask really changes)."""
    avail = sorted(os.sched_getaffinity(0))
    if len(avail) < 2:
        import pytest
        pytest.skip("one core")
    from vnect_amd import parallel as P
    root = str(tmp_path / "sys")
    lo, hi = avail[: len(avail) // 2], avail[len(avail) // 2:]
    _fake_sysfs(root, [(0, 0x05, 0, 0, P.format_cpulist(lo), 0), (0, 0x85, 0, 0, P.format_cpulist(hi), 1)])
    code = r"""
import os, sys, json
sys.path.insert(0, %r)
from vnect_amd import parallel as P
rep = P.bind_rank(0, enable=True, sysfs=%r)          # KFD order says device 0 is bus 0x05 ...
first = sorted(os.sched_getaffinity(0))
rep = P.rebind_by_bus(rep, 0x85, sysfs=%r)           # ... HIP says this rank's device sits on bus 0x85
same = P.rebind_by_bus(dict(rep), 0x85, sysfs=%r)    # (agreeing: nothing moves)
odd = P.rebind_by_bus({"bound": True, "bdf": "0000:05:00.0"}, 0x33, sysfs=%r)
print(json.dumps({"first": first, "now": sorted(os.sched_getaffinity(0)), "rep": rep, "same": same, "odd": odd}))
""" % (ROOT, root, root, root, root)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["first"] == lo and out["now"] == hi
    assert out["rep"]["rebound_after_hip_init"] is True and out["rep"]["bdf"] == "0000:85:00.0" and out["rep"]["bdf_is_the_hip_device"] is True
    assert out["rep"]["numa_node"] == 1 and out["rep"]["affinity"] == P.format_cpulist(hi)
    assert out["same"]["bdf_is_the_hip_device"] is True and out["same"]["affinity"] == P.format_cpulist(hi)
    assert out["odd"]["bdf_is_the_hip_device"] is False and "matching device" in out["odd"]["rebind"]
