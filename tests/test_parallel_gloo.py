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
