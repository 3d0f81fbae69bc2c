"""CPU: the C-ABI library loads, exports every symbol the header declares, and fails loudly without a GPU."""
import os
import re

import numpy as np
import pytest

from vnect_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "vnect_abi.h")).read()
    declared = set(re.findall(r"\b(vnect_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"vnect_handle", "vnect_config", "vnect_timings", "vnect_layer_info"}
    assert declared == set(_native.SYMBOLS), declared ^ set(_native.SYMBOLS)
    L = _native.lib()  # getattr on every symbol happens inside
    assert L.vnect_abi_version() == _native.ABI_VERSION
    # ... and NOTHING else: the library's dynamic symbol table is the C ABI (csrc/vnect.map) -- kernel launchers, runtime internals and the
    # kernels' host-side handles are local to it, so a host process cannot bind to (or collide with) anything but include/vnect_abi.h
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True)
    if out.returncode == 0:
        exported = {ln.split()[-1] for ln in out.stdout.splitlines() if ln.strip()}
        assert exported == set(_native.SYMBOLS), exported ^ set(_native.SYMBOLS)


def test_header_cites_reference_lines():
    hdr = open(os.path.join(ROOT, "include", "vnect_abi.h")).read()
    for fn in ("vnect_forward", "vnect_infer", "vnect_preprocess", "vnect_postprocess", "vnect_set_weight"):
        pos = hdr.index("int " + fn)
        assert re.search(r"src/\w+\.py:\d+", hdr[max(0, pos - 900):pos]), fn


def test_struct_sizes_match_header_layout():
    import ctypes as C
    assert C.sizeof(_native.Config) == 128
    assert _native.Config.keep_activations.offset == 112 and _native.Config.lanes.offset == 116
    assert _native.Config.preprocess_only.offset == 120 and _native.Config.exchange.offset == 124
    assert _native.Config.scales.offset == 16
    assert C.sizeof(_native.LayerInfo) == 64 + 7 * 4 + 4 + 16
    # vnect_timings: ABI v6 appended the two shader-clock doubles behind the v5 layout (56 bytes), which the library still accepts
    assert C.sizeof(_native.Timings) == 72 and _native.Timings.shader_cycles.offset == 56 and _native.Timings.conv_slot_ms.offset == 48
    hdr = open(os.path.join(ROOT, "include", "vnect_abi.h")).read()
    body = hdr[hdr.index("typedef struct vnect_timings {"):hdr.index("} vnect_timings;")]
    fields = re.findall(r"^\s*(?:int32_t|double)\s+(\w+);", body, re.M)
    assert fields == [k for k, _ in _native.Timings._fields_], fields


def test_shipped_library_was_built_with_every_probe_off():
    """The kernel sources carry timing probes behind -D switches, most of which give WRONG results on purpose (X3_DBG, WT_DBG, CH_DBG,
    POST_DBG, F32_NOSTORE, BF16_NOSTORE, VNECT_AB), tunables (NS_*, ARG_*) and the host runtime a test-only fault hook
    (VNECT_TEST_HOOKS).  vnect_build_info() reports what THIS binary was compiled with: the library the product loads -- the file
    bench.py and every GPU test run -- must have all of them at their product values, whatever EXTRA= a tuning session left behind."""
    info = _native.build_info()
    assert info["abi"] == str(_native.ABI_VERSION)
    assert info["probes_off"] == "1" and info["test_hooks"] == "0" and info["variant"] == "", info["text"]
    want = {"X3_DBG": "0", "WT_DBG": "0", "CH_DBG": "0", "F32_NOSTORE": "0", "BF16_NOSTORE": "0", "VNECT_AB": "0", "NS_6432": "5",
            "NS_32128": "5", "POST_DBG": "0", "ARG_SLABS": "8", "ARG_ROWSPLIT": "1"}
    got = dict(kv.split("=") for kv in (info["conv"] + " " + info["post"]).split())
    assert got == want, got
    assert "-O3" in info["flags"] and "gfx950" in info["flags"] and "-D" not in info["flags"], info["flags"]
    assert "clang" in info["compiler"].lower()
    # every probe macro the sources know is one the report covers (a new probe must be added to the report, or this fails)
    csrc = os.path.join(ROOT, "vnect_amd", "csrc")
    known = set()
    for f in ("conv.hip", "post.hip", "stem.hip"):
        known |= set(re.findall(r"^#ifndef ((?:[A-Z0-9]+_DBG|[A-Z0-9]+_NOSTORE|VNECT_AB|NS_[0-9]+|ARG_[A-Z]+))$", open(os.path.join(csrc, f)).read(), re.M))
    assert known == set(want), known ^ set(want)
    # the test twin says what it is, and is a different file from the one the product loads
    if os.path.exists(_native.TESTHOOKS_LIB):
        import ctypes as C
        T = C.CDLL(_native.TESTHOOKS_LIB)
        T.vnect_build_info.restype = C.c_char_p
        t = T.vnect_build_info().decode()
        assert "test_hooks=1" in t and "variant=_testhooks" in t and "probes_off=1" in t
    assert os.path.basename(_native.LIB_PATH) == "libvnect_hip.so" or os.environ.get("VNECT_LIB")


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_create_fails_loudly_without_gpu():
    with pytest.raises(_native.VnectError) as e:
        _native.Handle([1.0])
    assert e.value.code in (_native.E_NODEVICE, _native.E_HIP)


def test_bad_config_rejected():
    import ctypes as C
    L = _native.lib()
    cfg = _native.Config()
    cfg.struct_size = 7
    h = C.c_void_p()
    assert L.vnect_create(C.byref(cfg), C.byref(h)) == _native.E_ARG
    assert b"struct_size" in L.vnect_last_error(None)


def test_frame_validation():
    with pytest.raises(ValueError):
        _native._as_frame(np.zeros((4, 4, 3), np.float32))
    crop = np.zeros((100, 100, 3), np.uint8)[10:50, 20:60]  # non-contiguous rows are passed through by stride
    assert _native._as_frame(crop).strides[0] == 300
    # views the reference accepts but a (pointer, positive row stride) pair cannot describe are copied, not rejected
    img = np.arange(6 * 5 * 3, dtype=np.uint8).reshape(6, 5, 3)
    for view in (img[::-1], img[:, ::-1], img[..., ::-1], np.broadcast_to(img[:1], (6, 5, 3))):
        f = _native._as_frame(view)
        assert f.strides == (15, 3, 1) and np.array_equal(f, view)


def test_error_paths_return_codes_without_a_gpu():
    """Null handles / arguments come back as codes from every entry point (no abort, no exception), and the message of a failed
    vnect_create is per thread."""
    import ctypes as C
    import threading
    L = _native.lib()
    assert L.vnect_finalize(None) == _native.E_ARG
    assert L.vnect_set_weight(None, b"x", None, None, 1) == _native.E_ARG
    assert L.vnect_joint_filter(None, 2, None, 0, 1.0, None) == _native.E_ARG
    assert L.vnect_collect(None, None, None) == _native.E_ARG
    assert L.vnect_comm_unique_id(None) == _native.E_ARG
    L.vnect_destroy(None)
    seen = {}

    def worker():
        seen["before"] = L.vnect_last_error(None)
        cfg = _native.Config()
        cfg.struct_size = C.sizeof(_native.Config)
        cfg.num_scales = 99
        h = C.c_void_p()
        seen["rc"] = L.vnect_create(C.byref(cfg), C.byref(h))
        seen["after"] = L.vnect_last_error(None)

    cfg = _native.Config()
    cfg.struct_size = 7
    h = C.c_void_p()
    assert L.vnect_create(C.byref(cfg), C.byref(h)) == _native.E_ARG and h.value is None
    t = threading.Thread(target=worker)
    t.start(), t.join()
    assert seen["rc"] == _native.E_ARG and b"num_scales" in seen["after"] and seen["before"] == b""
    assert b"struct_size" in L.vnect_last_error(None)   # this thread's message is untouched by the other thread's failure


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under vnect_amd/ (Python, C++, HIP) may import, open or link it, and the product
    library's dynamic dependencies must not name it.  (bench.py's cpu_baseline leg and __graft_entry__.smoke() are the only
    callers outside tests/.)"""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "vnect_amd")
    offenders = []
    for d, _, files in os.walk(pkg):
        if os.path.basename(d) in ("lib", "__pycache__") or os.sep + "lib" + os.sep in d + os.sep:
            continue
        for f in files:
            if not f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                continue
            text = open(os.path.join(d, f), errors="replace").read()
            if re.search(r"^\s*(import|from)\s+oracle\b|libvnect_oracle|vnect_oracle\.h|oracle/_build|\bvo_[a-z_]+\(", text, re.M):
                offenders.append(os.path.relpath(os.path.join(d, f), root))
    assert not offenders, offenders
    so = os.path.join(pkg, "lib", "libvnect_hip.so")
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "NEEDED" in needed and "oracle" not in needed
    # there is no CPU fallback either: the loader raises if the library is missing instead of substituting anything
    src = open(os.path.join(pkg, "_native.py")).read()
    assert "raise" in src and "oracle" not in src


def test_rccl_copy_already_mapped_is_reused(tmp_path):
    """vnect_comm_library / load_rccl (rt_comm.cpp): a process that has a librccl.so mapped already -- torch.distributed's "nccl"
    backend maps torch's bundled copy -- gets THAT copy (RTLD_NOLOAD: no second RCCL build in the process, nothing RTLD_GLOBAL);
    a process without one gets the ROCm copy through the library's RUNPATH.  Symbol lookup only: no GPU needed.  The same on the
    GPU with real communicators: tests/test_gpu_multirank.py."""
    import json
    import subprocess
    import sys
    try:
        import torch
    except Exception:
        pytest.skip("no torch: no bundled librccl.so to map first")
    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if not os.path.exists(bundled):
        pytest.skip("torch ships no librccl.so")
    code = ("import ctypes, json, sys\nsys.path.insert(0, %r)\n"
            "if sys.argv[1] != '-': ctypes.CDLL(sys.argv[1])\n"
            "from vnect_amd import _native\np, r = _native.Handle.comm_library()\n"
            "print(json.dumps({'path': p, 'reused': r, 'mapped': sorted({l.split()[-1] for l in open('/proc/self/maps') if 'librccl' in l})}))" % ROOT)
    env = {k: v for k, v in os.environ.items() if k != "VNECT_RCCL_LIB"}

    def run(first):
        r = subprocess.run([sys.executable, "-c", code, first], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])

    a = run(bundled)
    assert a["reused"] is True and os.path.samefile(a["path"], bundled) and len(a["mapped"]) == 1, a
    b = run("-")
    assert b["reused"] is False and "librccl.so" in b["path"] and not os.path.samefile(b["path"], bundled) and len(b["mapped"]) == 1, b
    # an explicit choice wins over both
    r = subprocess.run([sys.executable, "-c", code, "-"], capture_output=True, text=True, env=dict(env, VNECT_RCCL_LIB=bundled), timeout=300)
    c = json.loads(r.stdout.strip().splitlines()[-1])
    assert os.path.samefile(c["path"], bundled) and c["reused"] is False, c
    # ... and one that cannot be honoured is an error (VNECT_E_COMM), not a reason to pick another copy
    r = subprocess.run([sys.executable, "-c", code, "-"], capture_output=True, text=True, env=dict(env, VNECT_RCCL_LIB="/nonexistent/librccl.so"), timeout=300)
    assert r.returncode != 0 and "librccl.so not available" in r.stderr and "code %d" % _native.E_COMM in r.stderr, r.stderr[-800:]


C_CALLER = r"""
#include <stdio.h>
#include <string.h>
#include "vnect_abi.h"
int main(void)
{
    vnect_config cfg;
    vnect_handle* h = 0;
    vnect_timings t;
    int rc;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = 7;                                  /* a wrong size is refused before any device is touched */
    rc = vnect_create(&cfg, &h);
    printf("abi %d\n", vnect_abi_version());
    printf("bad_create %d %s\n", rc, vnect_last_error(0));
    printf("build %s\n", vnect_build_info());
    t.struct_size = (int32_t)sizeof t;
    printf("timings_null %d\n", vnect_get_timings(0, &t));
    vnect_destroy(0);
    return h != 0;
}
"""


def test_a_plain_c_program_links_and_calls_the_abi(tmp_path):
    """The boundary is a C ABI, not a Python extension: include/vnect_abi.h compiles as pedantic C99 (and C++11), and a C program linked
    against libvnect_hip.so calls it with no Python, no torch and no GPU (it only exercises entry points that need none)."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "caller.c"
    src.write_text(C_CALLER)
    inc = os.path.join(ROOT, "include")
    libdir = os.path.dirname(_native.LIB_PATH)
    for cc, std, lang in (("gcc", "-std=c99", "c"), ("g++", "-std=c++11", "c++")):
        r = subprocess.run([cc, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", lang, "-I", inc, str(src)],
                           capture_output=True, text=True)
        assert r.returncode == 0, (cc, r.stderr[-1500:])
    exe = tmp_path / "caller"
    r = subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lvnect_hip", "-Wl,-rpath," + libdir,
                        "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr[-1500:])
    out = dict(ln.split(" ", 1) for ln in r.stdout.strip().splitlines())
    assert out["abi"] == str(_native.ABI_VERSION) and out["bad_create"].startswith("%d " % _native.E_ARG) and "struct_size" in out["bad_create"]
    assert "probes_off=1" in out["build"] and out["timings_null"] == str(_native.E_ARG)
