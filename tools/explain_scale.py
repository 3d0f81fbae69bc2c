"""Read N > 1 bench lines and say where a sub-linear curve comes from (CPU only; nobody is there to debug the driver's 8-GPU run).

    python3 tools/explain_scale.py SCALE_r06.json            # the driver's record (any JSON holding bench lines under "parsed" or in a list)
    python3 tools/explain_scale.py line_n1.json line_n8.json # or the lines themselves

For every line with `per_rank` (bench.py, round 6) it prints the ranks side by side -- own rate, p50 / p95, device time of a frame (HIP
events, profiling twin), conv stack, shader clock held under load, host gap per frame (1 / rate - device frame time), host binding -- and
an attribution of `value / N` against the job's own N = 1 loop (`n1_same_job`):
  clock   the conv stack got slower in step with the shader clock (eight GPUs share a node's power / cooling)
  device  the conv stack got slower at the same clock (memory / fabric contention between the ranks)
  host    the device frame time is unchanged, the gap between frames grew (launch path, CPU contention, a rank off its NUMA node)
  rank    one rank is slower than the others (the barrier-to-barrier `value` waits for it: the other ranks' `closing` column -- the closing
          barrier + synchronize of the timed region -- holds that wait)
Thresholds are 2 %: anything below is called "none"."""
import json
import sys


def lines_in(obj):
    if isinstance(obj, dict):
        if "per_rank" in obj and "value" in obj:
            yield obj
        for v in obj.values():
            yield from lines_in(v)
    elif isinstance(obj, list):
        for v in obj:
            yield from lines_in(v)


def explain(d):
    n = d["n_gpus"]
    pr = d["per_rank"]
    pyramid = d.get("scaling") == "strong"
    out = ["N = %d (%s, backend %s): value %.1f frames/s%s" % (n, "pyramid: one stream over %d GPUs" % n if pyramid else "stream replicas",
                                                              d.get("backend"), d["value"], "" if pyramid else " = %.1f per GPU" % (d["value"] / n))]
    out.append("  %4s %3s %9s %8s %8s %10s %9s %9s %9s %9s  %s" % ("rank", "dev", "frames/s", "p50 ms", "p95 ms", "device ms", "conv ms", "clock", "host gap", "closing", "host thread"))
    for r in pr:
        gap = 1e3 / r["frames_per_s"] - r["frame_ms_hip_events"]
        hb = r.get("host_binding", {})
        out.append("  %4d %3d %9.1f %8.4f %8.4f %10.4f %9.4f %9s %8.1fus %9s  %s" % (
            r["rank"], r["device"], r["frames_per_s"], r["latency_ms"]["p50"], r["latency_ms"]["p95"], r["frame_ms_hip_events"], r["conv_stack_ms"],
            ("%.0f" % r["shader_clock_mhz"]) if r.get("shader_clock_mhz") else "?", gap * 1e3,
            ("%.0fus" % r["closing_barrier_us"]) if r.get("closing_barrier_us") is not None else "?",
            ("cores %s (NUMA %s)" % (hb.get("affinity"), hb.get("numa_node"))) if hb.get("bound") else "unbound: %s" % hb.get("reason")))
    n1 = d.get("n1_same_job")
    verdicts = []
    rates = [r["frames_per_s"] for r in pr]
    if max(rates) > 1.02 * min(rates):
        slow = min(pr, key=lambda r: r["frames_per_s"])
        verdicts.append("rank: rank %d (device %d) runs %.1f %% below the fastest rank" % (slow["rank"], slow["device"], 100 * (1 - slow["frames_per_s"] / max(rates))))
    if n1 and not pyramid:
        per = d["value"] / n
        out.append("  same-job N = 1 loop (rank 0 alone): %.1f frames/s, p50 %.4f ms%s -> value / N is %.1f %% of it" % (
            n1["value"], n1["latency_ms"]["p50"], (", conv %.4f ms at %.0f MHz" % (n1["conv_stack_ms"], n1["shader_clock_mhz"])) if n1.get("shader_clock_mhz") else "",
            100 * per / n1["value"]))
        if n1.get("conv_stack_ms") and n1.get("shader_clock_mhz"):
            conv = sum(r["conv_stack_ms"] for r in pr) / n
            clk = sum(r["shader_clock_mhz"] for r in pr if r.get("shader_clock_mhz")) / max(1, sum(1 for r in pr if r.get("shader_clock_mhz")))
            dconv, dclk = conv / n1["conv_stack_ms"] - 1, 1 - clk / n1["shader_clock_mhz"]
            if dconv > 0.02 and dclk > 0.5 * dconv:
                verdicts.append("clock: conv stack +%.1f %% with the shader clock %.1f %% lower (%.0f -> %.0f MHz)" % (100 * dconv, 100 * dclk, n1["shader_clock_mhz"], clk))
            elif dconv > 0.02:
                verdicts.append("device: conv stack +%.1f %% at (almost) the same clock (%.0f -> %.0f MHz): contention between the ranks" % (100 * dconv, n1["shader_clock_mhz"], clk))
            gap1 = 1e3 / n1["value"] - n1["frame_ms_hip_events"]
            gapn = sum(1e3 / r["frames_per_s"] - r["frame_ms_hip_events"] for r in pr) / n
            if (gapn - gap1) > 0.02 * 1e3 / n1["value"]:
                verdicts.append("host: the gap between frames grew from %.1f to %.1f us per frame (launch path / CPU contention%s)" % (
                    gap1 * 1e3, gapn * 1e3, "" if all(r.get("host_binding", {}).get("bound") for r in pr) else "; not every rank is bound to its GPU's cores"))
    elif n1 and pyramid:
        out.append("  same-job N = 1 loop (the plain 3-scale handle on rank 0): %.1f frames/s -> the sharded job is %.2fx" % (n1["value"], d["value"] / n1["value"]))
    out.append("  attribution: " + ("; ".join(verdicts) if verdicts else "none (every rank within 2 % of the job's own N = 1 loop)"))
    return "\n".join(out)


def main(argv):
    seen = 0
    for f in argv:
        for d in lines_in(json.load(open(f))):
            print(explain(d))
            seen += 1
    if not seen:
        print("no bench line with `per_rank` in", argv)
    return 0 if seen else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
