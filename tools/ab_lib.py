"""A/B of two builds of the library on the same box: loads the given .so instead of the in-tree one, then runs a tools/
script with the remaining arguments (tools/perf_probe.py when none is named).
usage: python tools/ab_lib.py <path/to/libmiro_hip.so> [tools/<script>.py] [script args]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd"))
from miro_amd import binding  # noqa: E402

lib = sys.argv[1]
binding.load_library(os.path.abspath(lib))
print("library:", lib)
rest = sys.argv[2:]
script = os.path.join(ROOT, "tools", "perf_probe.py")
if rest and rest[0].endswith(".py"):
    script, rest = os.path.abspath(rest[0]), rest[1:]
sys.argv = [script] + rest
runpy.run_path(script, run_name="__main__")
