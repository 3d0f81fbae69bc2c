"""GPU (one MI355X): everything about the N > 1 paths (BASELINE.json configs[3] / configs[4]) that ONE GPU can prove.

The reference's seam: the S images of a frame's batch meet only in the merge (/root/reference/src/estimator.py:100-129), and one
estimator per video lives in a process of its own (run_estimator_ps.py:120-129).  What these tests pin down:

* torch.distributed's "nccl" backend (torch's BUNDLED librccl.so) and the library's own RCCL communicator (dlopen'ed) live in ONE
  process -- exactly what `bench.py --pyramid` and `parallel.PyramidJob` do -- and the library reuses the copy torch has mapped
  instead of loading /opt/rocm's second build of the same SONAME;
* the driver-shaped command `bench.py --pyramid --scales 1.0 --gpus 1` runs PyramidJob + ncclCommInitRank + one ncclAllGather per
  frame + the timed loop + the profile in one process under backend "nccl";
* `bench.py --gpus N` on a box with N - 1 devices fails at once, non-zero, naming the missing device.

What one GPU canNOT prove is listed in DESIGN.md section 6 (RCCL with 3 ranks, cross-device visibility of the p2p exchange's flags);
the tests that prove it on the first multi-GPU box are tests/test_gpu_multigpu.py (they skip on one device).
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


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT",
                                                             "VNECT_BENCH_BACKEND", "VNECT_BENCH_DEVICE", "VNECT_BENCH_WORKER")}
    env.update(extra)
    return env


TWO_RCCL_USERS = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "%d")
torch.cuda.set_device(0)
# 1. torch's own RCCL: a one-rank "nccl" process group and a real all-reduce on the GPU
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda") * 3
dist.all_reduce(t)
torch.cuda.synchronize()
assert t.tolist() == [3.0] * 4
mapped_before = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln})
# 2. the library's RCCL in the SAME process: a one-scale sharded handle (ncclCommInitRank + one ncclAllGather per frame)
from vnect_amd import _native
from vnect_amd.weights import synthetic_weights
from tests import helpers
w = synthetic_weights()
path, reused = _native.Handle.comm_library()
plain = _native.Handle([1.0]); plain.set_weights(w); plain.finalize()
shard = _native.Handle([1.0], pyramid=(0, 1)); shard.set_weights(w); shard.finalize()
uid = [_native.Handle.comm_unique_id()]
dist.broadcast_object_list(uid, src=0, device=torch.device("cuda", 0))      # as PyramidJob distributes it
shard.comm_init(0, 1, uid[0])
same = True
for k in range(4):
    frame = helpers.synth_frame(500 + k, smooth=True)
    tt = 1.7e9 + k / 30
    a2, a3 = plain.infer(frame, tt, tt + 0.001)
    b2, b3 = shard.infer(frame, tt, tt + 0.001)
    same = same and np.array_equal(a2, b2) and np.array_equal(a3, b3)
    dist.all_reduce(t)                                                        # torch's communicator keeps working in between
torch.cuda.synchronize()
mapped_after = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln})
plain.close(); shard.close()
dist.destroy_process_group()
print(json.dumps({"same": bool(same), "path": path, "reused": reused, "mapped_before": mapped_before, "mapped_after": mapped_after,
                  "torch_lib": os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "t": t.tolist()}))
"""


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def test_torch_rccl_and_library_rccl_in_one_process(tmp_path):
    """torch's "nccl" process group (its bundled RCCL) first, then a 1-scale sharded handle's comm_init + 4 frames in the same
    process: bit-equal to a plain handle, and the library REUSED torch's mapping -- one RCCL copy in the process, not two."""
    script = tmp_path / "two_rccl_users.py"
    script.write_text(TWO_RCCL_USERS % (ROOT, _free_port()))
    r = subprocess.run([sys.executable, str(script)], env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["same"], d
    assert d["t"] == [3.0] * 4                      # world size 1: the all-reduce is the identity, five times over
    assert len(d["mapped_before"]) == 1, d          # torch mapped exactly one RCCL ...
    assert d["mapped_after"] == d["mapped_before"], d   # ... and the library added no second copy
    assert d["reused"] is True and os.path.samefile(d["path"], d["mapped_before"][0]), d
    assert os.path.samefile(d["path"], d["torch_lib"]), d


def test_library_alone_opens_rocms_rccl(tmp_path):
    """Without torch in the process the library opens librccl.so.1 itself (RTLD_LOCAL, through its RUNPATH) and says so."""
    code = ("import sys; sys.path.insert(0, %r)\nfrom vnect_amd import _native\np, r = _native.Handle.comm_library()\n"
            "import json; print(json.dumps({'path': p, 'reused': r, 'torch': 'torch' in sys.modules}))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["reused"] is False and d["torch"] is False and "librccl.so" in d["path"] and "/torch/" not in d["path"], d


def test_bench_pyramid_one_rank_under_nccl():
    """`python bench.py --pyramid --scales 1.0 --gpus 1`: the configs[3] code path end to end with ONE rank -- backend "nccl", the
    process group built although the world is 1, PyramidJob, ncclAllGather in every frame, timed loop, profiling twin -- and the line
    says which RCCL it used and that it is a rehearsal."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--pyramid", "--scales", "1.0", "--gpus", "1", "--steps", "30",
                        "--warmup", "5", "--cpu-seconds", "0"], env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]   # ONE line on stdout: RCCL's version banner goes to stderr with the rest
    d = json.loads(lines[0])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "rehearsal_pyramid_rccl_one_rank.json"), "w") as f:
        json.dump(d, f, indent=1)
    assert d["backend"] == "nccl" and d["rccl_ranks"] == 1 and d["n_gpus"] == 1 and d["scaling"] == "strong"
    assert "rehearsal" in d["note"] and "REHEARSAL" in d["config"]["workload"] and "[1.0]" in d["config"]["workload"]
    assert d["exchange"].startswith("rccl") and "RCCL all-gather" in d["config"]["parallelism"]
    lib = d["rccl_library"]
    assert lib["reused_the_copy_already_mapped"] is True and lib["is_torchs_bundled_copy"] is True and len(lib["copies_mapped_in_this_process"]) == 1, lib
    assert 500 < d["value"] < 5000 and d["roofline"]["launches_per_frame"] >= 30 and 0.1 < d["roofline"]["frac"] < 1.0


def test_bench_more_ranks_than_devices_fails_fast():
    """`python bench.py --gpus N` (the driver's command shape) where only N - 1 devices exist -- `--gpus 2` on a one-GPU box, `--gpus 9`
    on an 8-GPU node: non-zero exit within seconds, the message names the missing device; nothing is left waiting in a rendezvous.
    Also under a launcher's environment (a rank whose LOCAL_RANK has no device).  Runs on every box (round 4 skipped it where a
    second device exists; the N > 1 parity tests proper are tests/test_gpu_multigpu.py)."""
    import torch
    have = torch.cuda.device_count()
    n = have + 1
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "1"], env=_clean_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    dt = time.time() - t0
    assert r.returncode != 0 and "HIP device %d is missing" % have in r.stderr and "exposes %d" % have in r.stderr, (r.returncode, r.stderr[-1500:])
    assert not r.stdout.strip()
    assert dt < 240, dt                             # one torch import in a throw-away child (minutes only on a box that has never imported torch), no rendezvous time-out
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "5", "--warmup", "1"],
                       env=_clean_env(RANK=str(have), LOCAL_RANK=str(have), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port())),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "needs HIP device %d" % have in r.stderr, (r.returncode, r.stderr[-1500:])
    assert time.time() - t0 < 240
