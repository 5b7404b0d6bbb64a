#!/usr/bin/env python3
"""Summarise a tools/profile_round.sh output directory: kernel-trace statistics of the step
kernel and HBM traffic per launch from the FETCH_SIZE / WRITE_SIZE passes.

gfx950 corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE / WRITE_SIZE are in KiB-units of
the L2's memory-side requests (x1024 -> bytes); FETCH_SIZE reads exactly 1/2 of the bytes of a
wide (16 B/lane) coalesced streaming read and is uncalibrated for other widths, WRITE_SIZE is
exact for 16 B/lane streaming stores.  This kernel mixes 16-byte traffic-block accesses with
4-byte per-env scalars, so the read side is CALIBRATED on a known byte count in the same access
pattern: at 4 194 304 envs (working set 1.5 GB >> 256 MB Infinity Cache) every input byte is
fetched from HBM exactly once, so  k_read = algorithmic_read_bytes / (FETCH_SIZE * 1024)  there,
and the same factor prices the headline size."""
import csv
import glob
import json
import os
import re
import statistics
import sys

d = sys.argv[1]
N, S = 8, 4


def kernel_sources_sha256():
    """The digest bench.py recomputes (bench.kernel_sources_sha256): which kernel sources the counter passes ran on."""
    import hashlib
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gym-acas2d_amd", "csrc")
    h = hashlib.sha256()
    for n in sorted(os.listdir(src)):
        if n.endswith((".hpp", ".inl", ".hip")) or n == "Makefile":
            h.update(n.encode() + b"\0" + open(os.path.join(src, n), "rb").read())
    return h.hexdigest()


def counter(name, E):
    vals = []
    files = glob.glob(os.path.join(d, "pmc_%s_%d" % (name, E), "**", "*counter_collection.csv"), recursive=True)
    files += glob.glob(os.path.join(d, "keep", "pmc_%s_%d_step_kernel.csv" % (name, E)))   # the extract profile_round.sh keeps
    for f in files[:1]:
        for r in csv.DictReader(open(f)):
            # the per-step kernel only: ROLLOUT / POLICY / SAMPLE all false (the last flag, ARENA, either way)
            if re.search(r"step_kernel<float, 4, 2, true, true, true, false, false, false(, (true|false))?>", r["Kernel_Name"]) \
                    and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"]))
    vals = vals[len(vals) // 4:]
    return statistics.mean(vals) if vals else None


def alg_bytes(E):
    rd = E * (S * (4 + 2 + 4 * N + 1) + 4)            # own x,y,psi,v + goal + traffic + action + steps
    wr = E * (S * (3 + 2 * N + (5 + 3 * N) + 1) + 4 + 1)
    return rd, wr


out = {"dir": os.path.basename(d)}
ks = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
if ks:
    rows = [r for r in csv.DictReader(open(ks[0])) if "step_kernel<float, 4, 2" in r["Name"]]
    rows.sort(key=lambda r: -int(r["Calls"]))          # per-step launches first, then the fused-rollout ones
    if rows:
        out["kernel_trace"] = {k: rows[0][k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "StdDev")}
    if len(rows) > 1:
        out["kernel_trace_rollout"] = {k: rows[1][k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "StdDev")}
for name in ("bench_traced", "bench_plain"):
    try:
        line = [l for l in open(os.path.join(d, name + ".json")) if l.startswith("{")][-1]
        j = json.loads(line)
        out[name] = {"value": j["value"], "launch_us": j["roofline"]["launch_us"], "frac": j["roofline"]["frac"]}
    except Exception as e:  # noqa: BLE001
        out[name] = str(e)
cal = {}
for E in (4194304, 65536):
    f, w = counter("FETCH_SIZE", E), counter("WRITE_SIZE", E)
    rd, wr = alg_bytes(E)
    cal[E] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "algorithmic_read_bytes": rd, "algorithmic_write_bytes": wr}
if cal[4194304]["FETCH_SIZE_KiB"]:
    k_read = cal[4194304]["algorithmic_read_bytes"] / (cal[4194304]["FETCH_SIZE_KiB"] * 1024)
    k_write = cal[4194304]["algorithmic_write_bytes"] / (cal[4194304]["WRITE_SIZE_KiB"] * 1024)
    out["calibration_4M_envs"] = {"k_read": k_read, "write_ratio_alg_over_counter": k_write, **cal[4194304]}
    if cal[65536]["FETCH_SIZE_KiB"]:
        rd_b = cal[65536]["FETCH_SIZE_KiB"] * 1024 * k_read
        wr_b = cal[65536]["WRITE_SIZE_KiB"] * 1024
        out["headline_65536_envs"] = {**cal[65536], "hbm_read_bytes_per_launch": rd_b,
                                      "hbm_write_bytes_per_launch": wr_b,
                                      "hbm_bytes_per_launch": rd_b + wr_b,
                                      "algorithmic_bytes_per_launch": sum(alg_bytes(65536)) }
# profiles/traffic.json (what bench.py prints as roofline.traffic): the read side under BOTH corrections -- the
# factor fitted at 4 M envs (every input byte fetched from HBM exactly once there) and the guide's exact x 2 for
# wide coalesced reads (MI355X_MICROARCH.md, HBM) -- for the headline size and for the 4 M-env run itself
if cal[4194304]["FETCH_SIZE_KiB"] and cal[65536]["FETCH_SIZE_KiB"]:
    recs = []
    for E in (65536, 4194304):
        f, w = cal[E]["FETCH_SIZE_KiB"] * 1024, cal[E]["WRITE_SIZE_KiB"] * 1024
        recs.append({"envs": E, "traffic": N, "dtype": "f32", "hbm_bytes_per_launch": int(round(f * k_read + w)),
                     "read_bytes": int(round(f * k_read)), "write_bytes": int(round(w)),
                     "read_correction_fitted_at_4M_envs": k_read,
                     "hbm_bytes_per_launch_guide_x2": int(round(2 * f + w)), "read_bytes_guide_x2": int(round(2 * f)),
                     "FETCH_SIZE_KiB": cal[E]["FETCH_SIZE_KiB"], "WRITE_SIZE_KiB": cal[E]["WRITE_SIZE_KiB"],
                     "algorithmic_bytes_per_launch": sum(alg_bytes(E)),
                     "build": os.environ.get("ACAS2D_BUILD_LABEL", "unlabelled"),
                     "kernel_sources_sha256": kernel_sources_sha256(),
                     "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --envs %d --launch "
                               "eager`, per-dispatch mean over the step kernel; tools/profile_round.sh" % E})
    out["traffic_json"] = recs
    json.dump(recs, open(os.path.join(d, "traffic.json"), "w"), indent=1)
# float64 builds: kernel-trace statistics and the SQ instruction mix per launch
for m in ("exact", "fast"):
    ks = glob.glob(os.path.join(d, "trace_f64_" + m, "**", "*kernel_stats.csv"), recursive=True)
    rec = {}
    if ks:
        rows = [r for r in csv.DictReader(open(ks[0])) if "step_kernel<double" in r["Name"]]
        rows.sort(key=lambda r: -int(r["Calls"]))
        if rows:
            rec["kernel_trace"] = {k: rows[0][k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs", "StdDev")}
    try:
        j = json.loads([l for l in open(os.path.join(d, "bench_f64_%s.json" % m)) if l.startswith("{")][-1])
        rec["bench_traced"] = {"value": j["value"], "launch_us": j["roofline"]["launch_us"], "frac": j["roofline"]["frac"]}
    except Exception as e:  # noqa: BLE001
        rec["bench_traced"] = str(e)
    out["f64_" + m] = rec
for tag in ("f32", "f64_exact", "f64_fast"):
    agg = {}
    for f in glob.glob(os.path.join(d, "pmc_sq_" + tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "step_kernel" in r["Kernel_Name"]:
                agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    if agg:
        out["sq_counters_per_launch_" + tag] = {k: statistics.mean(v[len(v) // 4:]) for k, v in sorted(agg.items())}
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
