#!/usr/bin/env python3
"""Condense the rocprofv3 passes of tools/collect_profiles.sh into the files that get committed under
profiles/: the kernel-stats table, the per-launch HBM traffic of the trace kernel (PMC FETCH_SIZE /
WRITE_SIZE, corrected as MI355X_MICROARCH.md section HBM prescribes) and a short markdown summary.

usage: summarize_profiles.py <gpurun_out/prof_TAG> <TAG>
"""
import csv
import glob
import json
import os
import re
import sys


def find(d, pat):
    f = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    return f[0] if f else None


def short(name):
    m = re.search(r"((?:trace|frame|trace_persistent)_kernel<[^>]*>|[a-z_]+_kernel)", name)
    return m.group(1) if m else name[:60]


def counter_table(path, counter):
    """kernel short name -> list of per-dispatch counter values"""
    out = {}
    if not path:
        return out
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != counter:
                continue
            out.setdefault(short(row["Kernel_Name"]), []).append(float(row["Counter_Value"]))
    return out


def main():
    d, tag = sys.argv[1], sys.argv[2]
    # on the GPU box only gpurun_out/ travels back: write next to the raw passes, copy into profiles/ afterwards
    pdir = sys.argv[3] if len(sys.argv) > 3 else os.path.join(d, "for_profiles")
    os.makedirs(pdir, exist_ok=True)
    stats = find(os.path.join(d, "trace"), "*kernel_stats.csv")
    bench_line = None
    for line in open(os.path.join(d, "trace.log"), errors="replace"):
        if line.startswith("{") and '"metric"' in line:
            bench_line = json.loads(line)
    rows = []
    if stats:
        with open(stats) as fh:
            rows = list(csv.DictReader(fh))
        with open(os.path.join(pdir, "%s_kernel_stats.csv" % tag), "w") as out:
            w = csv.writer(out)
            w.writerow(["kernel", "calls", "total_ms", "avg_ms", "percent", "min_ms", "max_ms"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6),
                            "%.4f" % (float(r["AverageNs"]) / 1e6), r["Percentage"],
                            "%.4f" % (float(r["MinNs"]) / 1e6), "%.4f" % (float(r["MaxNs"]) / 1e6)])
    fetch = counter_table(find(os.path.join(d, "pmc_fetch"), "*counter_collection.csv"), "FETCH_SIZE")
    write = counter_table(find(os.path.join(d, "pmc_write"), "*counter_collection.csv"), "WRITE_SIZE")

    md = ["# rocprofv3 summary `%s`" % tag, ""]
    if bench_line:
        md += ["bench line of the profiled run (profiled passes run at a lower clock; do not compare with un-profiled numbers):",
               "", "```json", json.dumps(bench_line), "```", ""]
    md += ["## kernel-trace --stats", "", "| kernel | calls | avg ms | total ms | % |", "|---|---|---|---|---|"]
    for r in rows[:12]:
        md.append("| `%s` | %s | %.4f | %.3f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e6,
                                                         float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
    md += ["", "## PMC passes (FETCH_SIZE / WRITE_SIZE, KiB per dispatch, averaged over dispatches)", "",
           "| kernel | dispatches | FETCH_SIZE KiB | WRITE_SIZE KiB |", "|---|---|---|---|"]
    kernels = sorted(set(fetch) | set(write))
    for k in kernels:
        f, w = fetch.get(k, []), write.get(k, [])
        md.append("| `%s` | %d | %s | %s |" % (k, max(len(f), len(w)),
                                               "%.1f" % (sum(f) / len(f)) if f else "-", "%.1f" % (sum(w) / len(w)) if w else "-"))

    traffic = None
    frame_k = [k for k in kernels if k.startswith("frame_kernel<")]
    if bench_line and frame_k and fetch.get(frame_k[0]) and write.get(frame_k[0]):
        # round 2: the whole step is one launch of frame_kernel; nothing streams through HBM but the 12-byte pixels
        k = frame_k[0]
        cfg = bench_line["config"]
        f = [x for x in fetch[k] if x > 0.5 * max(fetch[k])]
        w = [x for x in write[k] if x > 0.5 * max(write[k])]
        fetch_bytes = sum(f) / len(f) * 1024.0 * 2.0
        write_bytes = sum(w) / len(w) * 1024.0
        avg_ms = None
        for r in rows:
            if short(r["Name"]) == k:
                avg_ms = float(r["AverageNs"]) / 1e6
        traffic = {
            "workload": cfg["workload"], "tag": tag, "kernel": k,
            "fetch_size_kib_raw": sum(f) / len(f), "write_size_kib_raw": sum(w) / len(w), "fetch_correction": 2.0,
            "hbm_bytes_per_launch": fetch_bytes + write_bytes,
            "hbm_bytes_per_ray": (fetch_bytes + write_bytes) / cfg["rays_per_step"],
            "framebuffer_bytes_per_launch": 12.0 * cfg["rays_per_step"] / 2.0 / 64.0 if "64spp" in cfg["workload"] else None,
            "kernel_trace_avg_ms": avg_ms,
        }
        with open(os.path.join(pdir, "%s_traffic.json" % tag), "w") as out:
            json.dump(traffic, out, indent=1)
        md += ["", "## HBM traffic of the frame kernel", "",
               "FETCH_SIZE x 2 (gfx950 reports half of a wide coalesced read stream) + WRITE_SIZE per launch (= per step of "
               "%d rays): **%.1f MB**, %.3f B per ray -- the kernel keeps rays and hits in registers; what reaches HBM is the "
               "float framebuffer and the first touch of the 5 MB scene." % (cfg["rays_per_step"], traffic["hbm_bytes_per_launch"] / 1e6,
                                                                                traffic["hbm_bytes_per_ray"])]
        if avg_ms:
            md += ["", "kernel-trace average duration of `%s`: %.4f ms (the bench line's roofline.avg_launch_ms, from HIP events in the "
                   "same run: %s)" % (k, avg_ms, json.dumps(bench_line["roofline"].get("avg_launch_ms")))]
    main_k = [k for k in kernels if k.startswith("trace_kernel<true, false, false")]
    irr_k = [k for k in kernels if k.startswith("irradiance_kernel")]
    if bench_line and irr_k and "estimates_per_step" in bench_line.get("config", {}):
        # round 3, bench.py --config photon: the step is two launches of irradiance_kernel<false> (the counting build,
        # irradiance_kernel<true>, runs once per map after the timed region and shows as the slower pair of calls)
        k = irr_k[0]
        f = [x for x in fetch.get(k, []) if x > 0]
        w = [x for x in write.get(k, []) if x > 0]
        avg = [float(r["AverageNs"]) / 1e6 for r in rows if short(r["Name"]).startswith("irradiance_kernel")]
        calls = [int(r["Calls"]) for r in rows if short(r["Name"]).startswith("irradiance_kernel")]
        timed = avg[calls.index(max(calls))] if avg else None
        md += ["", "## irradiance_kernel", "",
               "kernel-trace average duration of the timed build (the row with the most calls): %s ms (the bench line's "
               "roofline.avg_launch_ms, from HIP events in the same run: %s)" % ("%.4f" % timed if timed else "-",
                                                                               json.dumps(bench_line["roofline"].get("avg_launch_ms")))]
        if f and w:
            fb, wb = sum(f) / len(f) * 1024.0 * 2.0, sum(w) / len(w) * 1024.0
            traffic = {"workload": bench_line["config"]["workload"], "tag": tag, "kernel": k, "fetch_size_kib_raw": sum(f) / len(f),
                       "write_size_kib_raw": sum(w) / len(w), "fetch_correction": 2.0, "hbm_bytes_per_launch": fb + wb,
                       "kernel_trace_avg_ms": timed}
            with open(os.path.join(pdir, "%s_traffic.json" % tag), "w") as out:
                json.dump(traffic, out, indent=1)
            md += ["", "HBM traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE, averaged over both builds' dispatches): **%.1f MB** -- the "
                   "9.6 MB of photon records are read from L2 / MALL, the queries (24 B) and results (12 B) stream." % ((fb + wb) / 1e6)]
    if not traffic and bench_line and main_k and fetch.get(main_k[0]) and write.get(main_k[0]) and "rays_per_step" in bench_line.get("config", {}):
        k = main_k[0]
        cfg = bench_line["config"]
        # the timed launches are the big ones; the STATS pre-pass uses another template instance
        f = [x for x in fetch[k] if x > 0.5 * max(fetch[k])]
        w = [x for x in write[k] if x > 0.5 * max(write[k])]
        rays_per_launch = cfg["rays_per_step"] / 2.0
        # calibration (MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950; WRITE_SIZE is
        # exact for 16 B/lane stores): the shadow-ray generator streams 16 B per hit in and 36 B per shadow ray out
        cal = None
        sk = [x for x in kernels if x.startswith("shadow_rays_kernel")]
        if sk and fetch.get(sk[0]):
            fs = max(fetch[sk[0]])
            cal = (rays_per_launch * 16.0) / (fs * 1024.0)
        fetch_bytes = sum(f) / len(f) * 1024.0 * 2.0
        write_bytes = sum(w) / len(w) * 1024.0
        traffic = {
            "workload": cfg["workload"], "tag": tag, "kernel": k,
            "fetch_size_kib_raw": sum(f) / len(f), "write_size_kib_raw": sum(w) / len(w),
            "fetch_correction": 2.0, "fetch_calibration_from_shadow_rays_kernel": cal,
            "hbm_bytes_per_launch": fetch_bytes + write_bytes,
            "hbm_bytes_per_ray": (fetch_bytes + write_bytes) / rays_per_launch,
            "algorithmic_stream_bytes_per_ray": 48.0,
        }
        with open(os.path.join(pdir, "%s_traffic.json" % tag), "w") as out:
            json.dump(traffic, out, indent=1)
        md += ["", "## HBM traffic of the trace kernel", "",
               "FETCH_SIZE x 2 (gfx950 reports half of a wide coalesced read stream; the shadow-ray generator, which "
               "streams a known 16 B per hit, gives a measured factor of %s) + WRITE_SIZE, per launch of ~%.0f M rays:" %
               ("%.2f" % cal if cal else "n/a", rays_per_launch / 1e6), "",
               "* %.1f MB per launch = **%.1f B per ray** (rays in + hits out are 48 B per ray; the node / triangle "
               "records are served by L2 / Infinity Cache)" % (traffic["hbm_bytes_per_launch"] / 1e6, traffic["hbm_bytes_per_ray"])]
    with open(os.path.join(pdir, "%s_summary.md" % tag), "w") as out:
        out.write("\n".join(md) + "\n")
    print("\n".join(md))
    if traffic:
        print(json.dumps(traffic))


if __name__ == "__main__":
    main()
