#!/usr/bin/env python3
"""Turns a tools/profile_gpu.sh run (gpurun_out/<tag>/) into the committed evidence under profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim)
  profiles/<tag>_bench.json         the bench line of the un-profiled run of the same command
  profiles/<tag>_summary.json       per-kernel launch averages from the trace and per-launch HBM traffic
                                    from the FETCH_SIZE / WRITE_SIZE passes
Traffic (MI355X_MICROARCH.md §HBM): FETCH_SIZE / WRITE_SIZE are in KiB per dispatch; on gfx950
FETCH_SIZE under-reports wide coalesced reads and other widths are uncalibrated, so the read side is
calibrated on this repository's own pack_scan_kernel, which reads every byte of a 96·N-byte buffer
exactly once with the same 8-byte-per-lane loads the iteration kernel uses: factor = 96·N / FETCH bytes.
usage: python tools/summarize_profile.py <tag> [n_points]
"""
import csv
import glob
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
n_points = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return hits[0] if hits else None


def short(name):
    for key in ("persistent_kernel", "iterate_kernel", "close_kernel", "pack_scan_kernel", "pack_arena_kernel", "upsert_kernel",
                "fold_rows_kernel", "table_clear_kernel"):
        if key in name:
            return key
    return name[:40]


sys.path.insert(0, ROOT)
from eskf_lio_amd import provenance  # noqa: E402

# what was measured: the library as built and the sources that decide the kernels' code (bench.py refuses to quote
# this summary's traffic for a library that matches neither)
summary = {"tag": tag, "library_sha256": provenance.library_sha256(), "kernel_source_sha256": provenance.kernel_source_sha256()}
stats = one("trace/**/*kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
trace = one("trace/**/*kernel_trace.csv")
if trace:
    dur = {}
    for r in csv.DictReader(open(trace)):
        dur.setdefault(short(r["Kernel_Name"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    summary["kernel_us"] = {k: {"calls": len(v), "mean": float(np.mean(v)) / 1e3, "median": float(np.median(v)) / 1e3,
                                "min": float(np.min(v)) / 1e3} for k, v in dur.items()}
    # rocprofv3's own reduction of the same trace (the *_kernel_stats.csv copied above) is the figure of record for the
    # mean: the two used to differ by 0.5 % (round 4's verdict) -- the trace rows and the stats rows are not the same
    # set when several instantiations share a short name, and the stats file sums per FULL kernel name
    if stats:
        by_short = {}
        for r in csv.DictReader(open(stats)):
            k = short(r["Name"])
            calls, total = int(r["Calls"]), float(r["TotalDurationNs"])
            acc = by_short.setdefault(k, [0, 0.0])
            acc[0] += calls
            acc[1] += total
        for k, (calls, total) in by_short.items():
            if k in summary["kernel_us"] and calls:
                summary["kernel_us"][k]["mean_from_trace_rows"] = summary["kernel_us"][k]["mean"]
                summary["kernel_us"][k]["mean"] = total / calls / 1e3
                summary["kernel_us"][k]["calls_in_stats_csv"] = calls


def counter(pass_dir, name):
    f = one(f"{pass_dir}/**/*counter_collection.csv")
    out = {}
    if not f:
        return out
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            out.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return out


fetch, write = counter("pmc_fetch", "FETCH_SIZE"), counter("pmc_write", "WRITE_SIZE")
if fetch:
    kib = 1024.0
    pack = np.mean(fetch.get("pack_scan_kernel", [np.nan])) * kib
    factor = 96.0 * n_points / pack if pack == pack and pack > 0 else float("nan")
    it_f = np.array(fetch.get("iterate_kernel", [np.nan])) * kib
    it_w = np.array(write.get("iterate_kernel", [np.nan])) * kib
    pk_f = np.array(fetch.get("persistent_kernel", [np.nan])) * kib
    pk_w = np.array(write.get("persistent_kernel", [np.nan])) * kib
    summary["traffic"] = {
        "unit": "bytes per launch",
        "persistent_kernel_fetch_raw": float(np.mean(pk_f)),
        "persistent_kernel_write": float(np.mean(pk_w)),
        "persistent_kernel_total_calibrated": float(np.mean(pk_f) * factor + np.mean(pk_w)),
        "iterate_kernel_fetch_raw": float(np.mean(it_f)),
        "iterate_kernel_write": float(np.mean(it_w)),
        "pack_scan_fetch_raw": float(pack), "pack_scan_known_read_bytes": 96.0 * n_points,
        "pack_scan_write": float(np.mean(write.get("pack_scan_kernel", [np.nan])) * kib),
        "read_calibration_factor": float(factor),
        "iterate_kernel_total_calibrated": float(np.mean(it_f) * factor + np.mean(it_w)),
        "iterate_kernel_total_guide_x2": float(np.mean(it_f) * 2.0 + np.mean(it_w)),
    }
plain = os.path.join(src, "bench_plain.json")
if os.path.exists(plain) and os.path.getsize(plain):
    shutil.copy(plain, os.path.join(dst, f"{tag}_bench.json"))
    try:
        summary["bench"] = json.loads(open(plain).read().strip().splitlines()[-1])
    except Exception:
        pass
json.dump(summary, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "bench"}, indent=1))
