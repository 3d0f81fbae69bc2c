"""Quantisation arithmetic of the fp32 launch plan: for every conv launch of a 3-scale frame, the matrix work it has to
spread over 1 024 SIMDs, what the plan in use makes of it, and the best any RECTANGULAR tile could do.  CPU only:

    python3 tools/tile_enum.py [profiles/r05_layer_table.txt]

Round 5 widens the search beyond the shapes the kernel has:
  * 16-row / 16-column granularity: v_mfma_f32_16x16x4_f32 has the rate of v_mfma_f32_32x32x2_f32 (2 048 FLOP per 32 cycles against 4 096 per 64), so a
    tile may be any R16 x C16 grid of 16 x 16 accumulators (4 registers each), not only 32 x 32 blocks -- non-power-of-two row counts included (112 x 16,
    80 x 32, 48 x 64 ...);
  * four OR eight consumer waves per workgroup (one or two per SIMD);
  * K groups 1 / 2 / 4 / 8 inside the workgroup.
Constraints kept: a wave holds at most 12 accumulators of 16 x 16 (= three 32 x 32: 48 registers, what the 64 x 96 x 2 shape uses); the tile's accumulators
divide evenly over (waves / K groups); a ring of >= 3 stages of (rows + columns) x 128 B x K groups plus the K-group partial sums fits 160 KB; launches that
carry a tail GEMM keep all N columns in one tile.

Model: a launch's tiles are dealt over 256 CUs; a CU with n tiles takes n x (tile's accumulator area / 4 SIMDs) of matrix time + n x TILE_US of per-tile
cost (epilogue, K-group reduction, ring refill: ~1.5 us measured for the streamed tiles of the plan in use, phase tables).  Unit = the whole K loop of ONE
32 x 32 block on one SIMD (K / 32 chunks x 1 024 cycles at 2.15 GHz).  Not modelled (both favour the plan in use): LDS fragment reads per FLOP (16 x 16 x 4
needs twice those of 32 x 32 x 2 for a square tile), LDS-DMA bytes per FLOP of narrow tiles.  The last column prices the best 16-granular plan against the
plan in use on the launch's MEASURED time, assuming 75 % of it is K loop -- an upper bound of what a new shape could buy.
"""
import glob
import math
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else (sorted(glob.glob("profiles/r0[0-9]_layer_table.txt")) or ["profiles/r04_layer_table.txt"])[-1]
rows = []
for ln in open(path):
    m = re.match(r"(\S+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)", ln)
    if m and int(m.group(2)) > 0:
        rows.append((m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)), int(m.group(7)), int(m.group(8)), float(m.group(9))))

LDS = 160 * 1024
TILE_US = 1.5


def best(Mrows, N, K, nphase, own_all_n, gran):
    """gran 32: the kernel's 32x32 accumulators, four consumer waves, K groups 1 / 2 / 4, <= 3 accumulators per wave (the round-4 search);
    gran 16: 16x16 accumulators, four or eight consumer waves, K groups 1 / 2 / 4 / 8, <= 12 accumulators per wave."""
    Mb, Nb = math.ceil(Mrows / gran), math.ceil(N / gran)
    unit = (gran / 32.0) ** 2  # accumulator area in 32x32 blocks
    out = []
    for R in range(1, (6 if gran == 32 else 12) + 1):
        for C in range(1, (8 if gran == 32 else 16) + 1):
            if own_all_n and C != Nb:
                continue
            for waves in ((4,) if gran == 32 else (4, 8)):
                for KG in ((1, 2, 4) if gran == 32 else (1, 2, 4, 8)):
                    if waves % KG or (R * C) % (waves // KG):
                        continue
                    acc_per_wave = R * C // (waves // KG)
                    if acc_per_wave > (3 if gran == 32 else 12):
                        continue
                    stage = (R + C) * gran * 128 * KG
                    part = (KG - 1) * R * C * int(4096 * unit)
                    if 3 * stage + part > LDS:
                        continue
                    tiles = math.ceil(Mb / R) * math.ceil(Nb / C) * nphase
                    per_cu = math.ceil(tiles / 256)
                    kloop_us = K / 32.0 * 1024 / 2150.0  # one 32x32 block's whole K loop on one SIMD
                    out.append((per_cu * (R * C * unit / 4.0 + TILE_US / kloop_us), tiles, R * gran, C * gran, KG, waves))
    return min(out) if out else None


print("layer table: %s" % path)
print("%-34s %6s %5s | %6s %6s | %-24s | %-28s | %s" % ("launch", "M", "N", "ideal", "in use", "best of 32x32 blocks", "best of 16x16 blocks", "us measured -> upper bound of the gain"))
tot_us = tot_gain = 0.0
for name, M, N, K, BM, BN, ks, wgs, us in rows:
    if name.startswith("conv1") or us <= 0:
        continue
    nphase = 4 if "deconv" in name else 1
    Mrows = M // nphase
    blocks = math.ceil(Mrows / 32) * math.ceil(N / 32) * nphase
    ideal = Mrows * N * nphase / (32.0 * 32.0) / 1024.0
    tail = ">" in name
    tiles = math.ceil(Mrows / BM) * math.ceil(N / BN) * nphase
    kloop_us = K / 32.0 * 1024 / 2150.0
    n_cu = math.ceil(tiles * max(ks, 1) / 256)
    inuse = n_cu * ((BM // 32) * (BN // 32) / 4.0 / max(ks, 1) + TILE_US / kloop_us)
    b32, b16 = best(Mrows, N, K, nphase, tail, 32), best(Mrows, N, K, nphase, tail, 16)
    gain = max(0.0, 1.0 - b16[0] / inuse) * 0.75 * us
    tot_us += us
    tot_gain += gain
    print("%-34s %6d %5d | %6.2f %6.2f | %3dx%-3d x%d %4d tiles %5.2f | %3dx%-3d x%d w%d %4d tiles %5.2f | %5.1f -> %4.1f%s" % (
        name[:34], Mrows, N, ideal, inuse, b32[2], b32[3], b32[4], b32[1], b32[0], b16[2], b16[3], b16[4], b16[5], b16[1], b16[0], us, gain,
        "  <--" if b16[0] < min(inuse, b32[0]) - 1e-9 else ""))
print("sum of the launches listed: %.1f us; upper bound of what 16-granular tiles could buy: %.1f us = %.1f %% of them (before any per-tile cost)" % (tot_us, tot_gain, 100 * tot_gain / tot_us))
