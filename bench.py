#!/usr/bin/env python3
"""bench.py -- frames/s of 368x368, 3-scale VNect inference on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A step is one pass of the hot path (VNectEstimator.__call__, /root/reference/src/estimator.py:97-142) over one
synthetic 368x368 BGR frame at scales [1.0, 0.8, 0.6], fp32 (BASELINE.json configs[1]).  Frames are resident in
HBM before the timed region; each step ends with the 21x2 + 21x3 joints back on the host (synchronous, as the
reference's tracking loop consumes them).  N>1 = N independent streams, one per GPU (weak scaling, no
collective on the data path; BASELINE.json configs[4]).  Weights are seeded synthetic (none ship with the reference).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCALES = [1.0, 0.8, 0.6]
FLOPS_PER_FRAME = 71.49e9      # 23.83 GFLOP per image x 3 scales (live graph, 2*MAC)
PEAK_FP32_MFMA = 157.3         # TFLOP/s, MI355X_MICROARCH.md chip table (v_mfma_f32_32x32x2_f32)
BF16_BYTES_PER_FRAME = 29.2e6 + 2 * 3 * 58.7e6   # see the roofline comment in main()
PEAK_BF16_MFMA = 2500.0        # TFLOP/s dense bf16 (same table; the 2:1-sparsity figure is not used)


def cpu_baseline(weights, budget_s):
    """The CPU oracle (a port of the reference path; TF1 itself cannot run here) on this box's host cores."""
    import oracle
    from tests import helpers
    est = oracle.OracleEstimator(weights=weights, scales=SCALES)
    frames = [helpers.synth_frame(1234 + k) for k in range(4)]
    # The port's blocked SGEMM stops scaling at ~16 threads on the 23x23 / 46x46 layers (measured on the GPU box:
    # 8 / 16 / 32 / 64 / 128 threads -> 3.6 / 4.0 / 3.7 / 2.9 / 1.6 frames/s), so it runs on at most 16 and says so.
    L = oracle.lib()
    L.vo_set_threads.argtypes = [__import__("ctypes").c_int]
    L.vo_set_threads(min(16, len(os.sched_getaffinity(0))))
    est(frames[0], 1.0, 1.0)  # warm-up (page-in, thread pool)
    n, t0 = 0, time.perf_counter()
    while True:
        est(frames[n % 4], 2.0 + n / 30, 2.0 + n / 30)
        n += 1
        dt = time.perf_counter() - t0
        if (dt >= budget_s and n >= 3) or n >= 400:
            break
    cores = oracle.lib().vo_sgemm_threads()
    return {"value": round(n / dt, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d frames of the same workload in %.1f s (C/OpenMP fp32 oracle, AVX2/AVX-512 SGEMM, threads capped at 16 "
                      "of %d host cores: more are slower)" % (n, dt, len(os.sched_getaffinity(0)))}


def cpu_framework_baseline(weights, budget_s):
    """A framework-grade stand-in for the reference's TF-CPU path (TF1 cannot run here): the same frames through
    torch-CPU (oneDNN) for the network, fp32, with the oracle's pre- and post-processing around it.  Reported beside
    `cpu_baseline`, labelled as a proxy; never a target."""
    import torch
    import oracle
    from tests import helpers, torch_net
    threads = min(16, len(os.sched_getaffinity(0)))
    torch.set_num_threads(threads)
    est = oracle.OracleEstimator(weights=weights, scales=SCALES)
    frames = [helpers.synth_frame(1234 + k) for k in range(4)]

    def frame(k, t):
        batch, scaler, off = oracle.gen_input_batch(frames[k % 4], SCALES)
        with torch.inference_mode():
            maps = torch_net.forward(weights, batch, dtype=torch.float32).numpy()
        return est.postprocess(maps, t, t, scaler, off[0], off[1])

    frame(0, 1.0)
    n, t0 = 0, time.perf_counter()
    while True:
        frame(n, 2.0 + n / 30)
        n += 1
        dt = time.perf_counter() - t0
        if (dt >= budget_s and n >= 3) or n >= 400:
            break
    return {"value": round(n / dt, 3), "unit": "frames/s", "cores": threads, "kind": "proxy: torch-CPU (oneDNN) fp32 network + oracle pre/post",
            "sample": "%d frames of the same workload in %.1f s on %d threads of %d host cores (torch %s)"
                      % (n, dt, threads, len(os.sched_getaffinity(0)), torch.__version__)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-aux", action="store_true",
                    help="skip the auxiliary legs (PCIe-inclusive, two frames in flight, two streams per GPU): they overlap "
                         "kernels by design, so a rocprofv3 run meant to describe the synchronous headline loop uses this")
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32",
                    help="fp32 = BASELINE.json configs[1] (default, the headline); bf16 = configs[2] (bf16 MFMA conv path)")
    ap.add_argument("--pyramid", action="store_true",
                    help="BASELINE.json configs[3]: ONE stream, one scale per GPU (needs --gpus 3), RCCL all-gather of the "
                         "maps; default for N>1 is N independent streams (configs[4])")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import numpy as np
    import torch

    from tests import helpers
    from vnect_amd import _native
    from vnect_amd.parallel import Group, aggregate_rate, stream_seed
    from vnect_amd.weights import synthetic_weights

    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    grp = Group("nccl")
    rank, local_rank = grp.rank, grp.local_rank

    weights = synthetic_weights()
    prec = _native.BF16 if args.precision == "bf16" else _native.FP32
    if args.pyramid:
        if args.gpus != len(SCALES):
            sys.exit("--pyramid shards the %d scales over %d GPUs: use --gpus %d" % (len(SCALES), len(SCALES), len(SCALES)))
        h = _native.Handle(SCALES, device=local_rank, num_frame_slots=8, pyramid=(rank, world), precision=prec)
    else:
        # lanes=3: the extra lanes only ever run frames submitted while others are in flight (the pipelined leg below)
        h = _native.Handle(SCALES, device=local_rank, use_graph=not args.no_graph, num_frame_slots=8, precision=prec, lanes=3)
    h.set_weights(weights)
    h.finalize()
    if args.pyramid:  # rank 0 makes the ncclUniqueId, torch.distributed carries it to the others
        import torch.distributed as dist
        uid = [_native.Handle.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        h.comm_init(rank, world, uid[0])
    # replicas: one synthetic video stream per rank, seeds 1234 + 1000*stream;
    # pyramid: every rank sees the SAME stream 0
    nslots = 8
    for k in range(nslots):
        h.upload_frame(k, helpers.synth_frame(stream_seed(0 if args.pyramid else rank, k)))

    t = 1.7e9
    for i in range(args.warmup):
        t += 1 / 30
        h.infer_resident(i % nslots, t, t + 1e-3)

    def barrier():
        grp.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    stamps = [t0]
    for i in range(args.steps):
        t += 1 / 30
        j2, j3 = h.infer_resident(i % nslots, t, t + 1e-3)
        stamps.append(time.perf_counter())  # per-frame latency distribution (median + p95)
    barrier()
    elapsed = time.perf_counter() - t0
    lat = np.diff(np.array(stamps)) * 1e3
    elapsed = grp.max_over_ranks(elapsed)
    assert np.all(np.isfinite(j2)) and np.all(np.isfinite(j3))

    # pipelined rate of the same stream (three frames in flight on three lanes: they overlap, only the filter kernels stay
    # ordered), reported beside the synchronous one
    pipelined = None
    if not args.no_aux:
        barrier()
        p0 = time.perf_counter()
        depth = 3  # frames in flight = lanes of the handle
        for i in range(args.steps):
            if i >= depth:
                h.collect()
            h.submit_resident(i % nslots, t + 1 + i / 30, t + 1 + i / 30 + 1e-3)
        for _ in range(min(depth, args.steps)):
            h.collect()
        torch.cuda.synchronize()
        pipelined = args.steps / (time.perf_counter() - p0)
    t += 1 + args.steps / 30 + 1

    # the same synchronous loop fed from HOST memory (vnect_infer: a pageable 406 KB frame crosses PCIe every step) --
    # the rate a caller of VNectEstimator.__call__ sees; never `value` (whose frames are resident in HBM)
    host_frames = [helpers.synth_frame(stream_seed(0 if args.pyramid else rank, k)) for k in range(nslots)]
    pcie = None
    if not args.pyramid and not args.no_aux:
        nh = max(args.steps // 3, 20)
        for i in range(5):
            t += 1 / 30
            h.infer(host_frames[i % nslots], t, t + 1e-3)
        torch.cuda.synchronize()
        q0 = time.perf_counter()
        for i in range(nh):
            t += 1 / 30
            h.infer(host_frames[i % nslots], t, t + 1e-3)
        pcie = nh / (time.perf_counter() - q0)
        for k in range(nslots):  # vnect_infer stages its frame in slot 0: restore the resident set
            h.upload_frame(k, host_frames[k])

    # two independent video streams sharing this GPU (two handles, two host threads): what the idle CUs between the
    # launches of one synchronous stream are worth.  Reported beside the headline, never as `value`.
    two_streams = None
    if not args.pyramid and rank == 0 and not args.no_aux:
        import threading
        h2 = _native.Handle(SCALES, device=local_rank, use_graph=not args.no_graph, num_frame_slots=8, precision=prec)
        h2.set_weights(weights)
        h2.finalize()
        for k in range(nslots):
            h2.upload_frame(k, helpers.synth_frame(stream_seed(rank + 1000, k)))

        def drive(hh, base, n):
            for i in range(n):
                hh.infer_resident(i % nslots, base + i / 30, base + i / 30 + 1e-3)

        drive(h2, t + 50, args.warmup)
        ths = [threading.Thread(target=drive, args=(hh, t + 100, args.steps)) for hh in (h, h2)]
        torch.cuda.synchronize()
        q0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        torch.cuda.synchronize()
        two_streams = 2 * args.steps / (time.perf_counter() - q0)
        h2.close()
        t += 100 + args.steps / 30 + 1

    out = None
    tim = None
    if rank == 0 or args.pyramid:  # pyramid: every inference is collective, so every rank must take part
        # Dominant kernel = vnect::conv_stream_kernel<BM,BN,KG,NS,BF,PROF> (the whole conv stack).
        # Its launch durations are taken live from a profiling twin of the frame graph in which every conv
        # kernel stamps its first-wave start and last-wave end with the 100 MHz device clock (what rocprofv3's
        # kernel trace reports); HIP events on the library's stream bracket the whole replayed frame.
        h.set_profiling(True)
        h.reset_timings()
        nprof = min(max(args.steps // 4, 10), 100)
        for i in range(nprof):
            t += 1 / 30
            h.infer_resident(i % nslots, t + 10, t + 10 + 1e-3)
        tim = h.timings()
        h.set_profiling(False)
    if rank == 0:
        # Per-launch duration = the SLOT a conv kernel occupies on the stream (its start to the next kernel's start): that is
        # what rocprofv3 reports per kernel (dispatch -> completion, durations abut) and what a launch costs the frame.  The
        # first-wave-start .. last-wave-end time of the same launches is carried beside it.
        conv_ms = tim["conv_slot_ms"] / nprof
        exec_ms = tim["conv_ms"] / nprof
        achieved = FLOPS_PER_FRAME / (len(SCALES) if args.pyramid else 1) / (conv_ms * 1e-3) / 1e12  # per GPU
        traffic = None
        tj = os.path.join(ROOT, "profiles", "traffic_latest.json" if args.precision == "fp32" else "r01_traffic_bf16.json")
        if os.path.exists(tj):
            traffic = json.load(open(tj)).get("hbm_bytes_per_frame")
        # rocprofv3's average for the same kernels (committed summary of the same command): its durations run from the
        # dispatch packet to the completion signal, i.e. ~1.5-2 us per launch more than first-wave-start .. last-wave-end
        rocprof_us = None
        rj = os.path.join(ROOT, "profiles", "r01_conv_roofline.json" if args.precision == "fp32" else "r01_conv_roofline_bf16.json")
        if os.path.exists(rj):
            rocprof_us = json.load(open(rj)).get("conv_avg_us_per_launch")
        peak = PEAK_FP32_MFMA if args.precision == "fp32" else PEAK_BF16_MFMA
        # what the vendor libraries need for the same layers (committed measurement of tools/vendor_ref.py, same box class)
        vendor = None
        vj = os.path.join(ROOT, "profiles", "r01_vendor_ref.json")
        if os.path.exists(vj):
            vendor = json.load(open(vj)).get(args.precision)
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": "frames/sec, 368x368 3-scale VNect inference",
            "value": round(aggregate_rate(1 if args.pyramid else args.gpus, args.steps, elapsed), 2),
            "unit": "frames/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong" if args.pyramid else "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": "368x368x3 uint8 BGR frame -> 21 joints; scales [1.0,0.8,0.6]; %s; batch 1 "
                                   "(BASELINE.json configs[%d]); N>1 = N independent streams, one per GPU"
                                   % (("fp32", 1) if args.precision == "fp32" else ("bf16 operands, fp32 accumulate", 2)),
                       "weights": "seeded synthetic (reference ships none)", "frames_resident_in_hbm": True,
                       "hip_graph": not args.no_graph and not args.pyramid, "sync_per_frame": True,
                       "parallelism": "pyramid: 1 scale per GPU + RCCL all-gather" if args.pyramid else "stream replicas"},
            "latency_ms": {"p50": round(float(np.percentile(lat, 50)), 4), "p95": round(float(np.percentile(lat, 95)), 4),
                           "max": round(float(lat.max()), 4)},
            "pcie_inclusive_frames_per_s_per_gpu": None if pcie is None else round(pcie, 2),
            "pipelined_frames_per_s_per_gpu": None if pipelined is None else round(pipelined, 2),
            "two_streams_on_one_gpu_frames_per_s": None if two_streams is None else round(two_streams, 2),
            "roofline": dict(
                         # fp32: the conv stack is MFMA-bound.  bf16: 16x the MFMA rate makes the
                         # same stack memory/latency-bound -- priced against HBM with its algorithmic bytes (bf16 weights
                         # 29.2 MB + every layer output written once and read once: 2 x 3 x 58.7 MB); the MFMA view is kept.
                         **({"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA, "unit": "TFLOP/s",
                             "frac": round(achieved / PEAK_FP32_MFMA, 4), "traffic": traffic} if args.precision == "fp32" else
                            {"bound": "hbm", "achieved": round(BF16_BYTES_PER_FRAME / (conv_ms * 1e-3) / 1e9, 1), "peak": 8000.0,
                             "unit": "GB/s", "frac": round(BF16_BYTES_PER_FRAME / (conv_ms * 1e-3) / 8e12, 4), "traffic": traffic,
                             "algorithmic_bytes_per_frame": BF16_BYTES_PER_FRAME,
                             "mfma_view": {"achieved_tflops": round(achieved, 2), "peak": PEAK_BF16_MFMA,
                                           "frac": round(achieved / PEAK_BF16_MFMA, 4)}}),
                         kernel="vnect::conv_stream_kernel<BM,BN,KG,NS,%s,0>: 64x64x1 (5-stage LDS ring), 64x32x2 and 32x32x4 "
                                "(in-workgroup K groups) on the 23x23 layers; %d launches per frame"
                                % ("false" if args.precision == "fp32" else "true", tim["conv_launches"]),
                         launches_per_frame=tim["conv_launches"],
                         avg_launch_us=round(conv_ms * 1e3 / tim["conv_launches"], 3),
                         first_to_last_wave={"avg_launch_us": round(exec_ms * 1e3 / tim["conv_launches"], 3),
                                             "achieved": round(FLOPS_PER_FRAME / (len(SCALES) if args.pyramid else 1) / (exec_ms * 1e-3) / 1e12, 3),
                                             "frac": round(FLOPS_PER_FRAME / (len(SCALES) if args.pyramid else 1) / (exec_ms * 1e-3) / 1e12 / peak, 4)},
                         rocprofv3_avg_launch_us=None if rocprof_us is None else round(rocprof_us, 3),
                         kernel_ms_per_frame=round(conv_ms, 4), flops_per_frame=FLOPS_PER_FRAME,
                         conv_stack_span_ms=round(tim["net_ms"] / nprof, 4),
                         hip_event_frame_ms=round(tim["total_ms"] / nprof, 4),
                         traffic_source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command, profiles/ (per frame)",
                         vendor_libraries_same_layers=vendor),
        }
    h.close()
    if rank == 0:
        if args.cpu_seconds > 0 and args.gpus == 1:
            out["cpu_baseline"] = cpu_baseline(weights, args.cpu_seconds)
            try:
                out["cpu_baseline_framework"] = cpu_framework_baseline(weights, min(args.cpu_seconds, 8.0))
            except Exception as e:  # noqa: the proxy is optional evidence, the bench line is not
                out["cpu_baseline_framework"] = {"error": str(e)[:200]}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    grp.close()


if __name__ == "__main__":
    main()
