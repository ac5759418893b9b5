"""A/B of two builds of the library on the same box: loads the given .so instead of the in-tree one, then runs
tools/perf_probe.py with the remaining arguments.  usage: python tools/ab_lib.py <path/to/libmiro_hip.so> [perf_probe args]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd"))
from miro_amd import binding  # noqa: E402

lib = sys.argv[1]
binding.load_library(os.path.abspath(lib))
print("library:", lib)
sys.argv = [os.path.join(ROOT, "tools", "perf_probe.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
