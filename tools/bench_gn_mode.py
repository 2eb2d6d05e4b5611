"""bench.py with a GroupNorm kernel-selection mode forced (dsc_debug_set_gn_mode): python tools/bench_gn_mode.py MODE [bench args]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionspatialcontrol_amd import _lib  # noqa: E402

mode = int(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
_lib.load_library().dsc_debug_set_gn_mode(mode)
import bench  # noqa: E402

bench.main()
