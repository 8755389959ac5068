"""bench.py against another build of the library (A/B on one device): GMLM_LIB=path/to/libgmlm_hip.so python tools/dev/ab_lib.py [bench flags]"""
import os
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, root)
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[1:]
if os.environ.get("GMLM_LIB"):
    import gmlm_amd._lib as _L
    _L.LIB_PATH = os.path.abspath(os.environ["GMLM_LIB"])
import bench
bench.main()
