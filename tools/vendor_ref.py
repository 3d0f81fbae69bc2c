"""What the vendor libraries reach on the SAME layers (calibration for DESIGN.md / profiles, never part of the product):
every conv-type layer of the VNect graph at batch 3 (scales [1.0, 0.8, 0.6] -> three 368x368 images), fp32,

  * as torch.nn.functional.conv2d / conv_transpose2d in channels_last (MIOpen picks the kernel), and
  * as the plain GEMM of its im2col shape, torch.matmul (rocBLAS / hipBLASLt) -- no gather, so an upper bound for any
    library-GEMM-based convolution,

timed with torch.cuda events on the GPU box, beside this repo's own per-layer times (tools/layer_table.py).
Usage (GPU box): python tools/vendor_ref.py [--bf16] [--budget SECONDS]
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ap = argparse.ArgumentParser()
ap.add_argument("--bf16", action="store_true")
ap.add_argument("--budget", type=float, default=600.0, help="stop starting new layers after this many seconds")
args = ap.parse_args()
dt = torch.bfloat16 if args.bf16 else torch.float32
torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False
dev = torch.device("cuda:0")
B = 3

# (name, Cin, Cout, k, stride, H_in, count) -- src/vnect_model.py:27-217; `count` identical layers in the live graph
L = [("conv1", 3, 64, 7, 2, 368, 1),
     ("res2a_branch1", 64, 256, 1, 1, 92, 1), ("res2x_branch2a_64", 64, 64, 1, 1, 92, 1), ("res2b_branch2a", 256, 64, 1, 1, 92, 1),
     ("res2x_branch2b", 64, 64, 3, 1, 92, 3), ("res2x_branch2c", 64, 256, 1, 1, 92, 3),
     ("res3a_branch1", 256, 512, 1, 2, 92, 1), ("res3a_branch2a", 256, 128, 1, 2, 92, 1), ("res3x_branch2a", 512, 128, 1, 1, 46, 3),
     ("res3x_branch2b", 128, 128, 3, 1, 46, 4), ("res3x_branch2c", 128, 512, 1, 1, 46, 4),
     ("res4a_branch1", 512, 1024, 1, 2, 46, 1), ("res4a_branch2a", 512, 256, 1, 2, 46, 1), ("res4x_branch2a", 1024, 256, 1, 1, 23, 5),
     ("res4x_branch2b", 256, 256, 3, 1, 23, 6), ("res4x_branch2c", 256, 1024, 1, 1, 23, 6),
     ("res5a_branch1", 1024, 1024, 1, 1, 23, 1), ("res5a_branch2a", 1024, 512, 1, 1, 23, 1), ("res5a_branch2b", 512, 512, 3, 1, 23, 1),
     ("res5a_branch2c", 512, 1024, 1, 1, 23, 1), ("res5b_branch2a", 1024, 256, 1, 1, 23, 1), ("res5b_branch2b", 256, 128, 3, 1, 23, 1),
     ("res5b_branch2c", 128, 256, 1, 1, 23, 1), ("res5c_deconv(63+128)", 256, 191, -4, 2, 23, 1),
     ("res5c_branch2b", 212, 128, 3, 1, 46, 1), ("res5c_branch2c", 128, 84, 1, 1, 46, 1)]


def timed(fn, iters=30):
    """us per call, device time only: `iters` calls replayed from one graph (eager torch calls are host-bound below ~18 us)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(side):
            fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=side):
                for _ in range(iters):
                    fn()
        run = g.replay
        per = iters
    except Exception as e:  # noqa: fall back to eager timing and say so
        print("  (graph capture failed: %s -- eager timing)" % str(e)[:60], flush=True)

        def run():
            for _ in range(iters):
                fn()
        per = iters
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * per) * 1e3  # us


t_start = time.time()
rows = []
tot_conv = tot_gemm = tot_flops = 0.0
print("%-24s %5s %5s %2s %2s %4s %2s %9s %9s %9s %8s %8s" % ("layer", "Cin", "Cout", "k", "s", "H", "n", "GFLOP", "conv us", "gemm us", "conv TF", "gemm TF"), flush=True)
for name, cin, cout, k, s, H, n in L:
    if time.time() - t_start > args.budget:
        print("budget reached, stopping", flush=True)
        break
    g = torch.Generator(device="cpu").manual_seed(1)
    x = (torch.rand((B, cin, H, H), generator=g) - 0.5).to(dev, dt).contiguous(memory_format=torch.channels_last)
    if k > 0:
        w = (torch.rand((cout, cin, k, k), generator=g) - 0.5).to(dev, dt).contiguous(memory_format=torch.channels_last)
        Ho = (H - 1) // s + 1 if k == 1 else (H + s - 1) // s
        pad = 0 if k == 1 else k // 2  # (TF SAME is asymmetric for conv1; symmetric padding has the same cost)
        conv = lambda: F.conv2d(x, w, None, stride=s, padding=pad)
        M, K = B * Ho * Ho, cin * k * k
    else:  # transposed 4x4 stride 2: 4 sub-pixel phases of 4 taps each
        w = (torch.rand((cin, cout, 4, 4), generator=g) - 0.5).to(dev, dt)
        conv = lambda: F.conv_transpose2d(x, w, None, stride=2, padding=1)
        M, K = B * (2 * H) * (2 * H), cin * 4
    flops = 2.0 * M * K * cout
    a = (torch.rand((M, K), generator=g) - 0.5).to(dev, dt)
    bm = (torch.rand((K, cout), generator=g) - 0.5).to(dev, dt)
    try:
        tc = timed(conv)
    except Exception as e:  # noqa
        print("  conv failed for %s: %s" % (name, str(e)[:80]), flush=True)
        tc = float("nan")
    tg = timed(lambda: torch.matmul(a, bm))
    rows.append(dict(name=name, cin=cin, cout=cout, k=k, stride=s, H=H, count=n, gflop=flops / 1e9, conv_us=tc, gemm_us=tg))
    tot_conv += n * tc
    tot_gemm += n * tg
    tot_flops += n * flops
    print("%-24s %5d %5d %2d %2d %4d %2d %9.3f %9.1f %9.1f %8.1f %8.1f" % (name, cin, cout, k, s, H, n, flops / 1e9, tc, tg, flops / tc / 1e6, flops / tg / 1e6), flush=True)
print("frame total (count-weighted): %.2f GFLOP; library conv %.1f us (%.1f TF/s); plain GEMMs %.1f us (%.1f TF/s)" % (
    tot_flops / 1e9, tot_conv, tot_flops / tot_conv / 1e6, tot_gemm, tot_flops / tot_gemm / 1e6), flush=True)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "vendor_ref%s.json" % ("_bf16" if args.bf16 else ""))
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(dict(dtype=str(dt), rows=rows, conv_us=tot_conv, gemm_us=tot_gemm, gflop=tot_flops / 1e9, torch=torch.__version__), open(out, "w"), indent=1)
