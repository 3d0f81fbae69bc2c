"""Is post_kernel bound by cold instruction fetch?  The same launch back to back (vnect_postprocess in a loop: nothing else runs on
the GPU in between, so the instruction cache stays warm) against its time inside a frame (profiles/*kernel_stats.csv).  Run as

    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/post_warm -- python3 $GRAFT_REPO_ROOT/tools/post_warm.py

(the interpreter itself after `--`: this file has no `#!/usr/bin/env` line on purpose -- under rocprofv3 the preloaded profiler has
initialised the GPU before the program starts, and every exec hop behind that (env, bash -c, a launcher) is refused on this pool);
compare the post_kernel average."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import helpers
from vnect_amd import _native
from vnect_amd.weights import synthetic_weights
h = _native.Handle([1.0, 0.8, 0.6])
h.set_weights(synthetic_weights()); h.finalize()
maps = helpers.synth_maps(5, 3)
for k in range(200):
    h.postprocess(maps, 1.0 + k / 30, 1.0 + k / 30)
h.close()
