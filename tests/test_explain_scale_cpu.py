"""CPU: tools/explain_scale.py -- the attribution a first N > 1 bench line gets with nobody there to debug it (clock / device / host /
slow rank), on synthetic lines shaped like bench.py's (`per_rank`, `n1_same_job`), and on the committed one-GPU rehearsal lines."""
import copy
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _line(n, rate=1140.0, conv=0.853, clock=2345.0, frame=0.871):
    pr = [{"rank": r, "device": r, "frames_per_s": rate, "own_elapsed_s": 300 / rate, "latency_ms": {"p50": 1e3 / rate, "p95": 1.01e3 / rate},
           "conv_stack_ms": conv, "frame_ms_hip_events": frame, "shader_clock_mhz": clock,
           "host_binding": {"bound": True, "affinity": "0-63", "n_cpus": 64, "numa_node": 0}} for r in range(n)]
    return {"value": n * rate, "n_gpus": n, "steps": 300, "scaling": "weak", "backend": "nccl", "per_rank": pr,
            "n1_same_job": {"value": 1140.0, "latency_ms": {"p50": 0.877}, "conv_stack_ms": 0.853, "frame_ms_hip_events": 0.871, "shader_clock_mhz": 2345.0}}


def test_attribution_rules():
    import explain_scale as E
    assert "attribution: none" in E.explain(_line(8))
    # eight GPUs sharing the node's power: clock 8 % down, conv stack 8 % up, rate follows
    d = _line(8, rate=1060.0, conv=0.853 * 1.08, clock=2345.0 * 0.92, frame=0.871 + 0.068)
    t = E.explain(d)
    assert "clock:" in t and "device:" not in t and "host:" not in t, t
    # the same conv slowdown at full clock: contention between the ranks
    d = _line(8, rate=1060.0, conv=0.853 * 1.08, clock=2340.0, frame=0.871 + 0.068)
    t = E.explain(d)
    assert "device:" in t and "clock:" not in t, t
    # the device is as fast as alone, the frames are further apart: the host
    d = _line(8, rate=1000.0)
    t = E.explain(d)
    assert "host:" in t and "clock:" not in t and "device:" not in t, t
    d["per_rank"][5]["host_binding"] = {"bound": False, "reason": "no local_cpulist"}
    assert "not every rank is bound" in E.explain(d)
    # one slow rank
    d = _line(4)
    d["per_rank"][2]["frames_per_s"] = 1000.0
    t = E.explain(d)
    assert "rank: rank 2 (device 2) runs 12.3 % below" in t, t
    # a pyramid line is compared with the plain 3-scale handle of the same job
    d = _line(3, rate=1500.0)
    d["scaling"], d["value"] = "strong", 1500.0
    assert "the sharded job is 1.32x" in E.explain(d)


def test_reads_the_committed_rehearsal_lines_and_a_driver_record(tmp_path):
    f = os.path.join(ROOT, "profiles", "r06_rehearsal_replicas_one_gpu.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "explain_scale.py"), f], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode == 0 and "N = 2 (stream replicas" in r.stdout and "same-job N = 1 loop" in r.stdout, r.stdout + r.stderr
    # the driver's SCALE record nests the lines (one per N) under "parsed"
    rec = {"runs": [{"n": n, "parsed": _line(n)} for n in (1, 2, 4, 8)], "skipped": False}
    p = tmp_path / "SCALE.json"
    p.write_text(json.dumps(rec))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "explain_scale.py"), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.count("attribution:") == 4, r.stdout
    p.write_text(json.dumps({"skipped": True}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "explain_scale.py"), str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode == 1 and "no bench line" in r.stdout
