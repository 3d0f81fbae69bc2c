"""Per-layer cost of the epilogue stores: tools/layer_table.py's output for the product and for a build whose stores are compiled out
(make VARIANT=_nst EXTRA=-DF32_NOSTORE=1, or -DBF16_NOSTORE=1), side by side.  CPU only:
    python3 tools/store_cost.py product.txt product_again.txt no_store.txt"""
import re
import sys


def rd(f):
    d = {}
    for ln in open(f):
        m = re.match(r"(\S+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)", ln)
        if m:
            d[m.group(1)] = float(m.group(9))
    return d


a, c, b = rd(sys.argv[1]), rd(sys.argv[2]), rd(sys.argv[3])
tot = 0
print("%-44s %8s %8s %8s %6s" % ("layer", "product", "again", "no store", "diff"))
for k in a:
    p = (a[k] + c.get(k, a[k])) / 2
    print("%-44s %8.1f %8.1f %8.1f %+6.1f" % (k[:44], a[k], c.get(k, 0), b.get(k, 0), b.get(k, 0) - p))
    tot += b.get(k, 0) - p
print("sum of differences %.1f us" % tot)
