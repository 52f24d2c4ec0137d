#!/usr/bin/env python3
"""bench.py with the named entry points of the library left out (timing experiment: results are wrong by construction).

    python tools/bench_skip.py svae_gru_fwd_range,svae_embed_bwd [bench.py arguments]

The switch lives here, in a tool, and not in the environment: `_lib.install_skip` is never called by the package."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

names = [x for x in sys.argv[1].split(",") if x and x != "none"]
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
from svae_cf_amd import _lib  # noqa: E402
if names:
    _lib.install_skip(names)
import bench  # noqa: E402
bench.main()
