"""Condense tools/pmc_set.sh's passes: per counter the mean over the counted launches of the probe's trace kernel
(all dispatches but the helper ones, which run under another kernel name), plus the derived figures the round's
DESIGN quotes: VALU issue fraction, lane utilisation, L1 accesses per ray, wait split."""
import csv
import glob
import os
import re
import sys


def main():
    d = sys.argv[1]
    probe = ""
    for f in sorted(glob.glob(os.path.join(d, "g*.log"))):
        for line in open(f, errors="replace"):
            if line.startswith("PMC_PROBE"):
                probe = line.strip()
    acc = {}
    for f in glob.glob(os.path.join(d, "g*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "trace_kernel" not in k and "trace_persistent" not in k and "frame_kernel" not in k:
                continue
            m = re.search(r"((trace|frame)_[a-z_]*kernel<[^>]*>)", k)
            name = m.group(1) if m else k[:60]
            acc.setdefault(name, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    print(probe)
    rays = None
    m = re.search(r"rays=(\d+)", probe)
    if m:
        rays = int(m.group(1))
    # the counted kernel is the one with the most dispatches
    for name, c in sorted(acc.items(), key=lambda kv: -max(len(v) for v in kv[1].values())):
        n = max(len(v) for v in c.values())
        print("== %s (%d dispatches per pass)" % (name, n))
        mean = {k: sum(v) / len(v) for k, v in c.items()}
        for k in sorted(mean):
            print("  %-32s %.5g" % (k, mean[k]))
        g = mean.get
        if g("SQ_INSTS_VALU") and g("SQ_BUSY_CYCLES"):
            # SQ_BUSY_CYCLES sums over the 32 SQs' shader engines... on gfx950 it reads XCD-count x busy cycles x 4 (one per SE);
            # r01 calibration: SQ_BUSY_CYCLES / 32 = kernel cycles.  A wave64 VALU instruction holds its SIMD-32 for 2 cycles.
            cyc = g("SQ_BUSY_CYCLES") / 32.0
            print("  derived: kernel cycles %.4g ; VALU issue fraction = INSTS_VALU*2/(1024 SIMDs * cycles) = %.3f" %
                  (cyc, g("SQ_INSTS_VALU") * 2.0 / (1024.0 * cyc)))
        if g("SQ_THREAD_CYCLES_VALU") and g("SQ_INSTS_VALU"):
            print("  derived: active lanes per VALU instruction = %.1f of 64" % (g("SQ_THREAD_CYCLES_VALU") / g("SQ_INSTS_VALU")))
        if g("SQ_WAVE_CYCLES"):
            wc = g("SQ_WAVE_CYCLES")
            print("  derived: wave-cycle split: issuing %.1f %%, issue-stalled %.1f %%, waiting (s_waitcnt) %.1f %%" %
                  (100 * g("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * g("SQ_WAIT_INST_ANY", 0) / wc, 100 * g("SQ_WAIT_ANY", 0) / wc))
        if rays and g("TCP_TOTAL_CACHE_ACCESSES_sum") is not None:
            print("  derived: vector-L1 accesses per ray %.2f ; L2 read requests per ray %.3f" %
                  (g("TCP_TOTAL_CACHE_ACCESSES_sum") / rays, g("TCP_TCC_READ_REQ_sum", 0) / rays))
        if rays and g("SQ_INSTS_VALU"):
            print("  derived: per 64 rays: VALU %.0f SALU %.0f SMEM %.0f VMEM_RD %.0f LDS %.0f branch %.0f" % tuple(
                g(k, 0) * 64.0 / rays for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH")))
        break


if __name__ == "__main__":
    main()
