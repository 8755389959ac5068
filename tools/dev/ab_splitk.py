"""A/B of the split-K slicing of the BERT weight gradients on ONE device: `old` = 16 slices everywhere, token count padded to a
multiple of 16 (round 3 until the slice sweep); `new` = nn._SLICES_MEASURED + padding to a multiple of 112 (the repository's
state).  usage: python tools/dev/ab_splitk.py old|new [bench.py flags]"""
import os
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, root)
mode = sys.argv[1]
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[2:]
import gmlm_amd.model as _m
import gmlm_amd.nn as _nn
if mode == "old":
    _nn._SLICES_MEASURED.clear()
    _m.SPLITK_ROW_QUANTUM = 16
import bench
bench.main()
