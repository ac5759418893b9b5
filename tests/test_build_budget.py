"""Register / scratch budget of the compiled kernels, held to what was verified on hardware.

History: in round 2 an intermediate build of the path-tracing level kernel (80-register cap, the generators' Newton loops
unrolled: 286 spilled SGPRs, 94 spilled VGPRs, 328 bytes of scratch per lane, 198 KB of code) faulted on the GPU.  Round 3
re-created that build and read its ISA ("The level-kernel fault"): nothing in it is structurally different from the
builds that run -- same opcodes, same spill mechanics, no whole-wave-mode spills, scratch offsets inside the frame, long
branches through a reserved register pair -- and a probe kernel with 224 ... 1040 bytes of scratch per lane runs correctly at
the same launch shape (profiles/r03_scratch_probe.log).  The mechanism was NOT found.  What can be held instead of a guessed
threshold: every kernel of the product build is recorded (tests/golden/kernel_budget.json, tools/kernel_budget.py --write)
from a tree whose whole GPU suite -- parity, fuzz, level and specular cases -- was green, and a later build may not leave that
verified envelope without being re-verified on the GPU and re-recorded: no more spilled VGPRs, no more scratch, no lower
occupancy, no kernel the record does not know."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_every_kernel_stays_inside_the_envelope_verified_on_the_gpu():
    import kernel_budget
    cur = kernel_budget.current()
    assert cur, "build the library first (__graft_entry__.build())"
    rec = json.load(open(kernel_budget.MANIFEST))["kernels"]
    unknown = sorted(set(cur) - set(rec))
    assert not unknown, "kernels without a verified record (run the GPU suite, then tools/kernel_budget.py --write): %s" % unknown[:5]
    for name, c in cur.items():
        r = rec[name]
        assert not c["dynamic_stack"], name
        assert c["vgprs_spilled"] <= r["vgprs_spilled"], (name, c, r)
        assert c["scratch_bytes_per_lane"] <= r["scratch_bytes_per_lane"], (name, c, r)
        assert c["waves_per_simd"] >= r["waves_per_simd"], (name, c, r)
    # the kernels that carry the bench and the parity suite exist under the names the record knows
    for needle in ("frame_kernelILi794ELi0ELb0", "trace_kernel", "level_kernelILi1818ELi2", "irradiance_kernelILb0", "children_kernelILb1"):
        assert any(needle in k for k in cur), needle


def test_traversal_kernels_keep_their_occupancy():
    import kernel_budget
    for name, c in kernel_budget.current().items():
        if "trace_kernel" in name or "frame_kernel" in name:
            assert c["waves_per_simd"] >= 6, (name, c)
        if "irradiance_kernelILb0" in name:
            assert c["waves_per_simd"] >= 5 and c["vgprs_spilled"] <= 4, (name, c)
