"""CPU oracle for the VNect path -- TEST INFRASTRUCTURE, not product code.

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this package.  It wraps ``oracle/_build/libvnect_oracle.so`` (built by ``oracle/Makefile`` from
``vnect_net.c`` + ``vnect_post.c``; see ``vnect_oracle.h`` for what each function restates).
"""
from .oracle import (Oracle, OracleEstimator, OneEuro, build, cvround, extract_2d, extract_3d,  # noqa: F401
                     gen_input_batch, hm_pt_interp, lib, merge_scales, resize)
