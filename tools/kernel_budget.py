"""Register / scratch budget of every kernel of the product build, from hipcc's -Rpass-analysis=kernel-resource-usage remarks
(written next to the objects by the Makefile).  `python tools/kernel_budget.py --write` records the current build as
tests/golden/kernel_budget.json -- to be done only for a tree whose GPU suite (pytest -m gpu, incl. the fuzz cases) is green:
tests/test_build_budget.py then holds every later build to what was verified on hardware."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAT = re.compile(r"Function Name: (\S+).*?TotalSGPRs: (\d+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Dynamic Stack: (\w+).*?"
                 r"Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+)", re.S)
MANIFEST = os.path.join(ROOT, "tests", "golden", "kernel_budget.json")


def current(build_dir=None):
    build_dir = build_dir or os.path.join(ROOT, "cse168-raytracer_amd", "build")
    out = {}
    for f in sorted(glob.glob(os.path.join(build_dir, "*.resource-usage.txt"))):
        unit = os.path.basename(f).replace(".resource-usage.txt", "")
        for name, sgprs, vgprs, scratch, dyn, occ, sspill, vspill in PAT.findall(open(f).read()):
            out["%s:%s" % (unit, name)] = {"vgprs": int(vgprs), "scratch_bytes_per_lane": int(scratch), "dynamic_stack": dyn == "True",
                                           "waves_per_simd": int(occ), "sgprs_spilled": int(sspill), "vgprs_spilled": int(vspill)}
    return out


if __name__ == "__main__":
    cur = current()
    if "--write" in sys.argv:
        json.dump({"note": "recorded from a build whose GPU suite was green; regenerate with tools/kernel_budget.py --write after "
                           "re-verifying on the GPU", "kernels": cur}, open(MANIFEST, "w"), indent=1, sort_keys=True)
        print("wrote %d kernels to %s" % (len(cur), MANIFEST))
    else:
        worst = sorted(cur.items(), key=lambda kv: -kv[1]["vgprs_spilled"])[:10]
        for k, v in worst:
            print(k[:110], v)
