"""Consistency of the documents with the tree (CPU): every profile file and every test DESIGN.md / README.md / NEGATIVE_RESULTS.md / profiles/README.md
name exists, DESIGN.md stays under 50 KB, and the ABI version the documents quote is the header's."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "README.md", "NEGATIVE_RESULTS.md", "INTEGRATION.md", os.path.join("profiles", "README.md")]


def _text(name):
    return open(os.path.join(ROOT, name)).read()


def test_profile_files_named_in_the_documents_exist():
    have = set(os.listdir(os.path.join(ROOT, "profiles")))
    missing = []
    for doc in DOCS:
        for m in re.finditer(r"`(?:profiles/)?(r0\d_[A-Za-z0-9_]+\.(?:json|csv|txt))`", _text(doc)):
            if m.group(1) not in have:
                missing.append((doc, m.group(1)))
    assert not missing, missing


def test_tests_named_in_design_exist():
    defined = set()
    for f in glob.glob(os.path.join(ROOT, "tests", "test_*.py")):
        defined.update(re.findall(r"^def (test_[A-Za-z0-9_]+)", open(f).read(), re.M))
    named = set(re.findall(r"`(test_[A-Za-z0-9_]+)`", _text("DESIGN.md")))
    # a trailing `*` in the text ("test_joint_filter_*") names a family: the pattern `test_x_` must prefix at least one test
    unknown = [n for n in named if n not in defined and not (n.endswith("_") and any(d.startswith(n) for d in defined))]
    assert not unknown, sorted(unknown)
    files = set(re.findall(r"`tests/(test_[a-z_]+\.py)`", _text("DESIGN.md") + _text("README.md")))
    assert all(os.path.exists(os.path.join(ROOT, "tests", f)) for f in files), files


def test_design_is_short_and_quotes_the_headers_abi_version():
    assert os.path.getsize(os.path.join(ROOT, "DESIGN.md")) <= 50 * 1024
    v = int(re.search(r"#define VNECT_ABI_VERSION (\d+)", _text(os.path.join("include", "vnect_abi.h"))).group(1))
    assert ("ABI v%d" % v) in _text("DESIGN.md") and ("ABI v%d" % v) in _text("INTEGRATION.md")
    from vnect_amd import _native
    assert _native.ABI_VERSION == v


def test_rates_quoted_in_integration_are_the_committed_bench_line():
    """INTEGRATION.md quotes frames/s figures in ONE table, every row naming the key of the committed bench line it comes from
    (tools/sync_docs_rates.py rewrites the numbers from the file): a figure that no longer matches the line it cites fails here -- round 5's
    text still carried round 4's numbers."""
    import json
    text = _text("INTEGRATION.md")
    m = re.search(r"committed as `(profiles/r\d\d_bench_line\.json)`", text)
    assert m, "the table of rates must name the bench line it quotes"
    line = json.load(open(os.path.join(ROOT, m.group(1))))
    newest = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.match(r"r\d\d_bench_line\.json$", f))[-1]
    assert m.group(1) == "profiles/" + newest, "INTEGRATION.md quotes %s, the newest committed line is %s" % (m.group(1), newest)
    rows = re.findall(r"^\|[^|]*\| `([a-z0-9_.]+)` \| ([0-9 .]+) \|$", text, re.M)
    assert len(rows) >= 10, rows
    for key, num in rows:
        v = line
        for k in key.split("."):
            v = v[k]
        assert abs(float(num.replace(" ", "")) - float(v)) <= 0.006, (key, num, v)
    # no other sentence of the file carries a bare frames/s figure of its own (percentages and microseconds are fine)
    body = text[:text.index("## Rates quoted in this file")]
    assert not re.search(r"\d \d{3}(?:\.\d+)? frames/s", body), re.findall(r"[^.]*\d \d{3}(?:\.\d+)? frames/s[^.]*", body)[:3]
