"""Register budget of the compiled kernels (hipcc -Rpass-analysis=kernel-resource-usage, written next to the objects by
the Makefile): the traversal kernels are pinned to an occupancy with amdgpu_waves_per_eu, and a change that pushes one of
them far past its cap shows up as dozens of spilled registers -- a path-tracing level kernel with 77-94 spilled VGPRs on top
of ~280 spilled SGPRs faulted on the GPU (round 2, after the generators' Newton loops were unrolled), while the same
kernel with <= 34 runs every test and fuzz campaign.  Keep every kernel on the safe side of that."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAT = re.compile(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Dynamic Stack: (\w+).*?"
                 r"Occupancy \[waves/SIMD\]: (\d+).*?VGPRs Spill: (\d+)", re.S)


def test_no_kernel_spills_more_than_a_few_dozen_registers():
    files = glob.glob(os.path.join(ROOT, "cse168-raytracer_amd", "build", "*.resource-usage.txt"))
    assert files, "build the library first (__graft_entry__.build())"
    seen = 0
    for f in files:
        for name, vgprs, scratch, dyn, occ, spill in PAT.findall(open(f).read()):
            seen += 1
            assert dyn == "False", name
            assert int(spill) <= 48, (name, spill)
            assert int(scratch) <= 256, (name, scratch)
            if "trace_kernel" in name or "frame_kernel" in name:
                assert int(occ) >= 6, (name, occ)
    assert seen >= 40
