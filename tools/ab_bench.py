"""bench.py's headline on an alternative build of the library (same box A/B): GCPX_LIB=path python tools/ab_bench.py"""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_gcp_amd import runtime as rt
if os.environ.get("GCPX_LIB"):
    rt.load_library(os.environ["GCPX_LIB"])
sys.argv = ["bench.py", "--no-extras", "--no-cpu-baseline"]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
