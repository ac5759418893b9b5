"""Would it pay to write a generator's children grouped by direction octant inside each 2048-ray chunk?  The bounce queue of the
atrium (Ray::random at every hit of a 1920x1080x4 frame) and of the bunny (1024x1024x16), traced as generated and with every
chunk's rays stably sorted by octant (chunk sizes 2048 and 16384): same rays, same hits (as a multiset), different order.
usage: python tools/octant_chunk_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cse168-raytracer_amd"))

import torch  # noqa: E402

import miro_amd  # noqa: E402
from miro_amd import binding, scenes  # noqa: E402


def timed(sc, rays, n, out, flags, reps=5):
    st = torch.cuda.current_stream()
    sc.trace_device(rays, n, out, flags, stream=st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        sc.trace_device(rays, n, out, flags, stream=st)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, W, H, spp in (("sponza", 1920, 1080, 4), ("bunny", 1024, 1024, 16)):
    d = scenes.SCENES[name]
    sc = miro_amd.Scene(0)
    scenes.populate(sc, d)
    sc.build(4)
    n0 = W * H * spp
    rays0 = torch.empty((n0, 8), dtype=torch.float32, device="cuda")
    hits0 = torch.empty((n0, 4), dtype=torch.float32, device="cuda")
    sc.gen_eye_rays(binding.make_camera(d["eye"], d["lookat"], d["up"], d["fov"]), W, H, rays0, spp=spp, jitter=True, tiled=True)
    sc.trace_device(rays0, n0, hits0)
    q = torch.empty((n0, 8), dtype=torch.float32, device="cuda")
    qw = torch.empty((n0, 3), dtype=torch.float32, device="cuda")
    qp = torch.empty(n0, dtype=torch.int32, device="cuda")
    qi = torch.empty(n0, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    sc.gen_path_rays(rays0, hits0, None, None, None, n0, q, qw, qp, qi, cnt, spp=spp, kinds=binding.MR_PATH_DIFFUSE)
    m = int(cnt.item())
    q = q[:m].contiguous()
    out = torch.empty((m, 4), dtype=torch.float32, device="cuda")
    print("%s: %d bounce rays" % (name, m))
    base = {}
    for fname, fl in (("default", 0), ("incoherent", miro_amd.MR_TRACE_INCOHERENT)):
        base[fname] = timed(sc, q, m, out, fl)
        print("  as generated        %-10s %7.3f ms  %6.2f Grays/s" % (fname, base[fname], m / base[fname] / 1e6))
    octant = ((q[:, 4] < 0).to(torch.int64) | ((q[:, 5] < 0).to(torch.int64) << 1) | ((q[:, 6] < 0).to(torch.int64) << 2))
    idx = torch.arange(m, device="cuda")
    for chunk in (2048, 16384, 1 << 20):
        key = (idx // chunk) * 8 + octant
        order = torch.sort(key, stable=True).indices
        qs = q[order].contiguous()
        for fname, fl in (("default", 0), ("incoherent", miro_amd.MR_TRACE_INCOHERENT)):
            ms = timed(sc, qs, m, out, fl)
            print("  octants per %-7d %-10s %7.3f ms  %6.2f Grays/s  (%+.1f %%)" % (chunk, fname, ms, m / ms / 1e6, 100.0 * (base[fname] / ms - 1.0)))
    # the product call: order kernel + gathering trace, hit buffer compared with the plain trace's
    ref = torch.empty((m, 4), dtype=torch.float32, device="cuda")
    sc.trace_device(q, m, ref, 0)
    order = torch.empty(m, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream()
    for lg in (11, 12, 13, 14):
        for fname, fl in (("default", 0), ("incoherent", miro_amd.MR_TRACE_INCOHERENT)):
            out.zero_()
            sc.trace_grouped(q, m, out, order, fl, chunk_log2=lg, stream=st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(5):
                sc.trace_grouped(q, m, out, order, fl, chunk_log2=lg, stream=st)
            e1.record(st)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            same = bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
            print("  mr_trace_grouped 2^%-2d  %-10s %7.3f ms  %6.2f Grays/s  (%+.1f %%)  %s" % (
                lg, fname, ms, m / ms / 1e6, 100.0 * (base[fname] / ms - 1.0), "same hit buffer" if same else "DIFFERENT HITS"))
    # the product call with the generator's octant bytes (made here with torch: what mr_gen_path_rays' d_out_octants holds)
    octs = ((q[:, 4] < 0).to(torch.uint8) | ((q[:, 5] < 0).to(torch.uint8) << 1) | ((q[:, 6] < 0).to(torch.uint8) << 2)).contiguous()
    for lg in (13, 14):
        for fname, fl in (("default", 0), ("incoherent", miro_amd.MR_TRACE_INCOHERENT)):
            out.zero_()
            sc.trace_grouped(q, m, out, order, fl, chunk_log2=lg, stream=st, d_octants=octs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(5):
                sc.trace_grouped(q, m, out, order, fl, chunk_log2=lg, stream=st, d_octants=octs)
            e1.record(st)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            same = bool(torch.equal(out.view(torch.int32), ref.view(torch.int32)))
            print("  grouped, octant bytes 2^%-2d  %-10s %7.3f ms  %6.2f Grays/s  (%+.1f %%)  %s" % (
                lg, fname, ms, m / ms / 1e6, 100.0 * (base[fname] / ms - 1.0), "same hit buffer" if same else "DIFFERENT HITS"))
    # the gathering trace alone (the order of the last call reused): what the order kernel costs is the difference
    for fname, fl in (("default", 0), ("incoherent", miro_amd.MR_TRACE_INCOHERENT)):
        sc.trace_grouped(q, m, out, order, fl, chunk_log2=14, stream=st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            sc.trace_grouped(q, m, out, order, fl, chunk_log2=14 | miro_amd.MR_ORDER_GIVEN, stream=st)
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("  gather only      2^14  %-10s %7.3f ms  %6.2f Grays/s  (%+.1f %%)" % (fname, ms, m / ms / 1e6, 100.0 * (base[fname] / ms - 1.0)))
    del sc
