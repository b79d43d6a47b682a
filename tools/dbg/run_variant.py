import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nimblephysics_amd._lib as _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "dbg", sys.argv[1])
sys.argv = ["bench.py", "--steps", "16", "--warmup", "2", "--no-cpu-baseline"]
import runpy
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
