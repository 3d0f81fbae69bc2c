"""Quantisation arithmetic of the fp32 launch plan: for every conv launch of a 3-scale frame, the 32x32 blocks x K
it has to spread over 1 024 SIMDs, what the plan in use makes of it, and the best any RECTANGULAR tile could do -- R x C blocks of 32x32, 1 / 2 / 4
in-workgroup K groups over the four consumer waves, at most three accumulators per wave, an LDS ring that fits (>= 3 stages in 160 KB with the
K-group partial sums; two workgroups per CU where two rings fit).  Per-SIMD makespan in units of ONE block's whole K loop; CPU only:

    python3 tools/tile_enum.py [profiles/r04_layer_table.txt]

Model: a launch's tiles are dealt over 256 CUs; a CU with n tiles takes n x (R C / 4) block-K-loops per SIMD (co-resident workgroups share the
SIMDs, streamed ones follow each other: the same sum).  Not modelled: per-tile overheads (they favour FEWER, larger tiles), launches that
carry a tail GEMM (their tile must own all N columns) are restricted to C = N / 32.
"""
import math
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r04_layer_table.txt"
rows = []
for ln in open(path):
    m = re.match(r"(\S+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)", ln)
    if m and int(m.group(2)) > 0:
        rows.append((m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)), int(m.group(7)), int(m.group(8)), float(m.group(9))))


def best(Mrows, N, nphase, own_all_n):
    Mb, Nb = math.ceil(Mrows / 32), math.ceil(N / 32)
    out = []
    for R in range(1, 7):
        for C in range(1, 9):
            if own_all_n and C != Nb:
                continue
            for KG in (1, 2, 4):
                if (R * C * KG) % 4 or (R * C * KG) // 4 > 3 * KG:      # accumulators per wave = R C KG / 4 ... of which KG-fold K split: R C / (4 / KG) <= 3
                    continue
                if R * C // (4 // KG) > 3 if KG < 4 else R * C > 3:
                    continue
                stage = (R + C) * 32 * 128 * KG
                part = (KG - 1) * R * C * 4096
                if 3 * stage + part > 160 * 1024:
                    continue
                two = 2 * (5 * stage if KG == 1 else 3 * stage + part) <= 160 * 1024
                tiles = math.ceil(Mb / R) * math.ceil(Nb / C) * nphase
                per_cu = math.ceil(tiles / 256)
                out.append((per_cu * R * C / 4.0, tiles, R * 32, C * 32, KG, two))
    return min(out) if out else None


print("%-34s %6s %5s | %7s %7s | %-22s | %s" % ("launch", "M", "N", "ideal", "in use", "best rectangular tile", "makespan"))
for name, M, N, K, BM, BN, ks, wgs, us in rows:
    if name.startswith("conv1"):
        continue
    nphase = 4 if "deconv" in name else 1
    Mrows = M // nphase
    blocks = math.ceil(Mrows / 32) * math.ceil(N / 32) * nphase
    ideal = blocks / 1024.0
    tail = ">" in name
    kg = 4 // ((BM // 32) * (BN // 32)) if (BM // 32) * (BN // 32) <= 4 else 2
    tiles = math.ceil(Mrows / BM) * math.ceil(N / BN) * nphase
    inuse = math.ceil(tiles * max(ks, 1) / 256) * (BM // 32) * (BN // 32) / 4.0 / max(ks, 1)
    b = best(Mrows, N, nphase, tail)
    print("%-34s %6d %5d | %7.2f %7.2f | %3dx%-3d x%d %4d tiles%s | %.2f%s" % (name[:34], Mrows, N, ideal, inuse, b[2], b[3], b[4], b[1], " 2/CU" if b[5] else "     ",
                                                                      b[0], "   <-- better than the plan in use" if b[0] < inuse - 1e-9 else ""))
