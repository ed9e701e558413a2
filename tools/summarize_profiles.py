#!/usr/bin/env python3
"""gpurun_out/ (written by tools/collect_profiles.sh on the GPU box) -> profiles/ (committed).

    python tools/summarize_profiles.py r04

What it writes:
  <tag>_bench_all.json            the contract line of the PROFILED process (the one the kernel statistics belong to)
  <tag>_bench_detail.json         that process's full record
  <tag>_bench_all_plain.json      the line of the unprofiled run of the same command (carries the in-run HBM traffic of every config)
  <tag>_bench_detail_plain.json   its full record
  <tag>_process_kernel_stats.csv  rocprofv3 --stats of the WHOLE profiled process, every kernel and every size (pre-warm, placement survey, PCIe legs, extras):
                                  not the headline kernel's statistic -- that is the next file
  <tag>_timed_kernel_stats.csv    per kind: the dispatches of the TIMED region of that same process (from the kernel trace and the launch
                                  log bench.py wrote), beside the kernel_ms the line reports -- these two must agree
  <tag>_sq_summary.csv            SQ counters of the compute-bound kernels (separate --pmc runs)
"""
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT, PROF = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"


def find(pattern):
    hits = sorted(glob.glob(os.path.join(OUT, pattern), recursive=True))
    return hits[-1] if hits else None


def last_json_line(path):
    if not (path and os.path.exists(path)):
        return None
    rows = [ln for ln in open(path, errors="replace").read().splitlines() if ln.startswith("{")]
    return rows[-1] if rows else None


for src, dst in (("bench_all_profiled.json", f"{tag}_bench_all.json"), ("bench_all.json", f"{tag}_bench_all_plain.json")):
    line = last_json_line(os.path.join(OUT, src))
    if line:
        json.loads(line)
        open(os.path.join(PROF, dst), "w").write(line + "\n")
for src, dst in (("bench_detail_profiled.json", f"{tag}_bench_detail.json"), ("bench_detail_plain.json", f"{tag}_bench_detail_plain.json")):
    p = os.path.join(OUT, src)
    if os.path.exists(p):
        open(os.path.join(PROF, dst), "w").write(json.dumps(json.load(open(p)), indent=1) + "\n")

f = find("prof_all/**/*kernel_stats.csv")
if f:
    open(os.path.join(PROF, f"{tag}_process_kernel_stats.csv"), "w").write(open(f).read())

# the timed dispatches of the profiled process
trace, log = find("prof_all/**/*kernel_trace.csv"), os.path.join(OUT, "prof_all_launches.json")
if trace and os.path.exists(log):
    rows = []
    with open(trace) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows.sort()
    pos, table = 0, []
    TRACE_NAME = {"k_dct_bsh": "k_dct"}   # reported name -> the symbol in the trace (k_dct<LOG2M, FAM, true>)
    for ent in json.load(open(log)):
        picked = []   # (segment, duration)
        segs = ent.get("segments") or [["setup", ent["launches"]]]
        for seg, n in segs:
            got = 0
            while got < n and pos < len(rows):
                if TRACE_NAME.get(ent["kernel"], ent["kernel"]) in rows[pos][1]:
                    picked.append((seg, rows[pos][2]))
                    got += 1
                pos += 1
        timed = [d for s, d in picked if s == "timed"]
        if ent["kind"] and timed:
            table.append([ent["kind"], ent["kernel"], len(timed), round(statistics.mean(timed)), min(timed), round(statistics.median(timed)), max(timed),
                          round(ent["kernel_ms"] * 1e6), round(statistics.mean(timed) / (ent["kernel_ms"] * 1e6), 4)])
    with open(os.path.join(PROF, f"{tag}_timed_kernel_stats.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kind", "kernel", "timed_dispatches", "rocprof_mean_ns", "min_ns", "median_ns", "max_ns", "line_kernel_ns_hip_events", "rocprof_over_line"])
        w.writerows(table)


# SQ passes: matrix-core busy time, issue and LDS behaviour of the dominant kernel (mean per dispatch)
def all_counters(kind, leg):
    f = find(f"pmc_{kind}_{leg}/**/*counter_collection.csv")
    if not f:
        return {}
    acc = {}
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            s = acc.setdefault(name, {}).setdefault(r["Counter_Name"], [0, 0.0])
            s[0] += 1
            s[1] += float(r["Counter_Value"])
    return {k: {c: v / n for c, (n, v) in d.items()} for k, d in acc.items()}


KERNEL_OF = {"mel": "k_mel", "mfcc": "k_mel", "mel_mfcc": "k_mel", "cqt": "k_cqt", "stft": "k_stft_ft16", "dct": "k_dct", "mel64": "k_mel_ft8_f64", "cqt64": "k_cqt_ft_f64", "istft8192": "k_istft_ft8q", "imdct8192": "k_imdct_q"}
sq_rows = []
for kind, kname in KERNEL_OF.items():
    merged, label = {}, kname
    for leg in ("SQ", "LDS"):
        tab = all_counters(kind, leg)
        kern = next((k for k in tab if kname in k), None)
        if kern:
            merged.update(tab[kern])
            label = kern.split("<")[0].split("(")[0].split("::")[-1].strip() or kname   # the kernel that ran (k_mel2 for config 3 since round 4)
    if not merged:
        continue
    grbm, mfma = merged.get("GRBM_GUI_ACTIVE"), merged.get("SQ_VALU_MFMA_BUSY_CYCLES")
    derived = {}
    if grbm and mfma is not None:
        # rocprofv3's MfmaUtil: MFMA busy cycles (summed over the 1024 SIMDs) / (kernel cycles x SIMDs); GRBM_GUI_ACTIVE arrives
        # summed over the 8 XCDs
        derived["mfma_util"] = mfma / (grbm / 8.0 * 1024.0)
        derived["kernel_cycles"] = grbm / 8.0
    if merged.get("SQ_LDS_IDX_ACTIVE"):
        derived["lds_bank_conflict_over_active"] = merged.get("SQ_LDS_BANK_CONFLICT", 0.0) / merged["SQ_LDS_IDX_ACTIVE"]
    if merged.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
            if merged.get(c) is not None:
                derived[c.lower()[3:] + "_over_wave_cycles"] = merged[c] / merged["SQ_WAVE_CYCLES"]
    for c, v in sorted(merged.items()):
        sq_rows.append(f'{kind},"{label}",{c},{v:.1f}')
    for c, v in sorted(derived.items()):
        sq_rows.append(f'{kind},"{label}",{c},{v:.5f}')
if sq_rows:
    with open(os.path.join(PROF, f"{tag}_sq_summary.csv"), "w") as fh:
        fh.write("kind,kernel,counter_or_ratio,mean_per_dispatch\n" + "\n".join(sq_rows) + "\n")
print("profiles/ refreshed for", tag)
