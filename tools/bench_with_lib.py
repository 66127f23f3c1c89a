"""bench.py against another build of the library (A/B sessions): MMD_AMD_LIB=<path to .so> python tools/bench_with_lib.py [bench args]"""
import os
import runpy
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmd_amd import _lib
if os.environ.get("MMD_AMD_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MMD_AMD_LIB"])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
