"""Where do the idle VALU cycles of the bench frame sit?  Needs the measurement build of mr_frame.hip:
    make -C cse168-raytracer_amd VARIANT=_wgt FRAME_DEFS=-DMIRO_WG_TIMES
    MIRO_LIB=cse168-raytracer_amd/lib_wgt/libmiro_hip.so python tools/wg_timeline.py
Every workgroup of frame_kernel leaves its start / end time (100 MHz clock), HW_ID and XCC_ID; this prints how many workgroups
were resident over the launch (the tail), when each XCD and CU finished, and how evenly the work was spread."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import miro_amd  # noqa: E402
from miro_amd import binding, scenes  # noqa: E402
from miro_amd import frame as mframe  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    a = ap.parse_args()
    L = binding.lib()
    if not hasattr(L, "mr_debug_wg_times"):
        raise SystemExit("this library was not built with -DMIRO_WG_TIMES")
    d = scenes.SCENES[a.scene]
    sc = miro_amd.Scene(0)
    scenes.populate(sc, d)
    sc.build(4)
    fr = mframe.FusedFrame(sc, d, a.w, a.h, spp=a.spp)
    for _ in range(3):
        fr.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fr.step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    buf = np.zeros(4 * 131072, np.uint64)
    rc = L.mr_debug_wg_times(buf.ctypes.data_as(C.c_void_p), C.c_uint(4 * 131072))
    assert rc == 0, rc
    t = buf.reshape(-1, 4)
    t = t[t[:, 0] > 0]                                    # the workgroups of the launch (the buffer starts zeroed)
    n_wg = len(t)
    t0, t1 = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
    hw, xcc = t[:, 2].astype(np.int64), t[:, 3].astype(np.int64) & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    start, end = t0.min(), t1.max()
    span = float(end - start)
    print("%s %dx%dx%d: launch %.3f ms by events, %.3f ms first start -> last end (100 MHz ticks), %d workgroups on %d CUs of %d XCDs" % (
        a.scene, a.w, a.h, a.spp, ms, span / 1e5, n_wg, len(np.unique(key)), len(np.unique(xcc))))
    dur = (t1 - t0).astype(np.float64)
    print("workgroup lifetimes: mean %.1f us, median %.1f, p5 %.1f, p95 %.1f, max %.1f" % tuple(
        x / 100.0 for x in (dur.mean(), np.median(dur), np.percentile(dur, 5), np.percentile(dur, 95), dur.max())))
    # resident workgroups over time
    ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t1, -np.ones_like(t1)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    res = np.cumsum(ev[:, 1])
    tt = (ev[:, 0] - start) / span
    full = res.max()
    print("resident workgroups: peak %d (%.2f per CU)" % (full, full / len(np.unique(key))))
    for lo, hi in ((0, .02), (.02, .1), (.1, .5), (.5, .8), (.8, .9), (.9, .95), (.95, .98), (.98, 1.0)):
        m = (tt >= lo) & (tt < hi)
        if m.any():
            # time-weighted mean residency in the window
            seg_t = np.diff(np.concatenate([ev[m, 0], ev[m, 0][-1:]])).astype(np.float64)
            w = (res[m] * seg_t).sum() / max(seg_t.sum(), 1.0)
            print("  %4.0f-%3.0f %% of the launch: %.0f resident on average (%.2f of peak)" % (lo * 100, hi * 100, w, w / full))
    area = float(((res[:-1]) * np.diff(ev[:, 0])).sum())
    print("time-weighted residency over the launch: %.3f of peak" % (area / (full * span)))
    # per XCD / CU
    for name, k in (("XCD", xcc), ("CU", key)):
        ends = np.array([t1[k == v].max() - start for v in np.unique(k)], np.float64) / span
        busy = np.array([dur[k == v].sum() for v in np.unique(k)])
        cnt = np.array([(k == v).sum() for v in np.unique(k)])
        print("%s: last end at %.3f .. %.3f of the launch (mean %.3f); workgroups per %s %d .. %d; summed lifetimes min/mean/max %.2f / %.2f / %.2f ms" % (
            name, ends.min(), ends.max(), ends.mean(), name, cnt.min(), cnt.max(), busy.min() / 1e5, busy.mean() / 1e5, busy.max() / 1e5))
    ts = np.sort(t0) - start
    print("the launch fills: workgroup 256 / 896 / 1792 / 3584 (by start time) starts %.1f / %.1f / %.1f / %.1f us after the first" % tuple(
        ts[min(k, n_wg - 1)] / 100.0 for k in (255, 895, 1791, 3583)))
    # the last workgroups to start: how long they lived (graded schedule: the short classes)
    order = np.argsort(t0)
    for lo, hi in ((0, n_wg - 8000), (n_wg - 8000, n_wg - 4000), (n_wg - 4000, n_wg - 2000), (n_wg - 2000, n_wg)):
        sel = order[max(lo, 0):hi]
        if len(sel):
            print("  workgroups %6d..%6d by start time: start at %.3f..%.3f of the launch, mean lifetime %.1f us" % (
                max(lo, 0), hi, (t0[sel].min() - start) / span, (t0[sel].max() - start) / span, dur[sel].mean() / 100.0))
    starts_late = ((t0 - start) / span)
    print("workgroup starts: %.1f %% in the first 2 %% of the launch, last start at %.3f" % (100.0 * (starts_late < 0.02).mean(), starts_late.max()))


if __name__ == "__main__":
    main()
