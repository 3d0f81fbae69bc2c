"""Rewrite the numbers of INTEGRATION.md's table of rates from the committed bench line it cites (CPU; run after a new profiles/rNN_bench_line.json
is committed).  The table names a key of the line per row; tests/test_docs_cpu.py checks the same correspondence."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROW = re.compile(r"^(\|[^|]*\| `([a-z0-9_.]+)` \| )([^|]*)( \|)$")


def lookup(d, path):
    for k in path.split("."):
        d = d[k]
    return d


def fmt(x):
    s = ("%.2f" % x).rstrip("0").rstrip(".")
    whole, _, frac = s.partition(".")
    if len(whole) > 3:
        whole = whole[:-3] + " " + whole[-3:]
    return whole + ("." + frac if frac else "")


def main():
    path = os.path.join(ROOT, "INTEGRATION.md")
    text = open(path).read()
    m = re.search(r"committed as `(profiles/r\d\d_bench_line\.json)`", text)
    line = json.load(open(os.path.join(ROOT, m.group(1))))
    out, n = [], 0
    for ln in text.split("\n"):
        r = ROW.match(ln)
        if r:
            ln = r.group(1) + fmt(lookup(line, r.group(2))) + r.group(4)
            n += 1
        out.append(ln)
    open(path, "w").write("\n".join(out))
    print("rewrote %d rates from %s" % (n, m.group(1)))


if __name__ == "__main__":
    sys.exit(main())
