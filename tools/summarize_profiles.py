#!/usr/bin/env python3
"""gpurun_out/ (written by tools/collect_profiles.sh on the GPU box) -> profiles/ (committed).

    python tools/summarize_profiles.py r02
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT, PROF = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
KINDS = ["stft", "istft", "mdct", "imdct", "mel", "mfcc", "cqt", "dct", "stft_offgrid", "stft4096", "stft4096_h1024", "istft4096", "mdct4096"]


def find(pattern):
    hits = sorted(glob.glob(os.path.join(OUT, pattern), recursive=True))
    return hits[-1] if hits else None


def stats_rows(kind):
    f = find(f"prof_{kind}/**/*kernel_stats.csv")
    if not f:
        return None, []
    with open(f) as fh:
        rows = list(csv.reader(fh))
    return rows[0], [r for r in rows[1:] if r and "zafx::" in r[0]]


# bench lines
for k in KINDS + ["stft_tf", "all"]:
    src = os.path.join(OUT, f"bench_{k}.json")
    if os.path.exists(src) and os.path.getsize(src):
        line = open(src).read().strip().splitlines()[-1]
        json.loads(line)
        open(os.path.join(PROF, f"{tag}_bench_{k}.json"), "w").write(line + "\n")

# the bench lines the PROFILED runs printed themselves (prof_<kind>.log): the kernel statistics below belong to these processes, and
# where an allocation lands differs from process to process -- compare rocprofv3's mean with THIS line's kernel_ms
prof_lines = {}
for k in KINDS:
    log = os.path.join(OUT, f"prof_{k}.log")
    if os.path.exists(log):
        rows = [ln for ln in open(log, errors="replace").read().splitlines() if ln.startswith("{")]
        if rows:
            try:
                line = json.loads(rows[-1])
                prof_lines[k] = {"steps": line["steps"], "warmup": line["warmup"], "ms_per_step": line["ms_per_step"],
                                 "kernel_ms": line["roofline"]["kernel_ms"], "kernel_ms_median": line["roofline"]["kernel_ms_median"],
                                 "kernel": line["roofline"]["kernel"]}
            except (ValueError, KeyError):
                pass
if prof_lines:
    open(os.path.join(PROF, f"{tag}_profiled_runs.json"), "w").write(json.dumps(prof_lines, indent=1) + "\n")

# kernel statistics: the headline run in full, the dominant rows of the others in one file
f = find("prof_stft/**/*kernel_stats.csv")
if f:
    open(os.path.join(PROF, f"{tag}_stft_kernel_stats.csv"), "w").write(open(f).read())
with open(os.path.join(PROF, f"{tag}_other_kernel_stats.csv"), "w", newline="") as fh:
    w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
    wrote_header = False
    for k in KINDS[1:]:
        header, rows = stats_rows(k)
        if header and not wrote_header:
            w.writerow(["kind"] + header)
            wrote_header = True
        for r in rows:
            w.writerow([k] + r)


# PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs) of the dominant kernel of every transform
def counter_means(kind, counter):
    f = find(f"pmc_{kind}_{counter}/**/*counter_collection.csv")
    if not f:
        return {}
    acc = {}
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].split("(")[0]
            s = acc.setdefault(name, [0, 0.0])
            s[0] += 1
            s[1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in acc.items()}


rows = []
for kind in KINDS:
    fetch, write = counter_means(kind, "FETCH_SIZE"), counter_means(kind, "WRITE_SIZE")
    bench_file = os.path.join(PROF, f"{tag}_bench_{kind}.json")
    if not (fetch and write and os.path.exists(bench_file)):
        continue
    bench = json.loads(open(bench_file).read())
    kname = bench["roofline"]["kernel"]
    kern = next((k for k in fetch if kname in k), None)
    if kern is None or kern not in write:
        continue
    for cname, tab in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        for k, (n, v) in sorted(tab.items()):
            if k == kern or (kind == "stft" and "copyBuffer" in k):
                rows.append(f'{kind},"{k}",{cname},{n},{v:.3f}')
    copy = next((k for k in fetch if "copyBuffer" in k), None)
    doc = {
        "kernel": kname,
        "fetch_size_kb_raw": fetch[kern][1],
        "write_size_kb_raw": write[kern][1],
        "fetch_correction": 2.0,
        "write_correction": 1.0,
        "hbm_bytes_per_launch": fetch[kern][1] * 1024 * 2.0 + write[kern][1] * 1024,
        "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
        "collected": "%s: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --kind %s --steps 3`" % (tag, kind),
        "note": "FETCH_SIZE/WRITE_SIZE in KB from separate rocprofv3 --pmc passes (profiles/%s_pmc_summary.csv); gfx950 "
                "FETCH_SIZE counts half the bytes of streaming reads (MI355X_MICROARCH.md, HBM section; confirmed on the "
                "device-to-device copy of the stft run), WRITE_SIZE is exact on that copy" % tag,
    }
    if kind == "stft" and copy:
        copy_bytes = 441000 * 4 * 8   # bench.py replicates blocks of 8 clips device-to-device
        doc["calibration"] = {"kernel": copy, "bytes_per_dispatch": copy_bytes,
                              "fetch_true_over_counter": copy_bytes / (fetch[copy][1] * 1024),
                              "write_true_over_counter": copy_bytes / (write[copy][1] * 1024)}
    open(os.path.join(PROF, f"pmc_{kind}.json"), "w").write(json.dumps(doc, indent=1) + "\n")
if rows:
    with open(os.path.join(PROF, f"{tag}_pmc_summary.csv"), "w") as fh:
        fh.write("kind,kernel,counter,dispatches,mean_value_KB\n" + "\n".join(rows) + "\n")
# the bench lines were produced before this run's PMC summary existed: carry the fresh traffic figure into them
for kind in KINDS:
    pmc, bench_file = os.path.join(PROF, f"pmc_{kind}.json"), os.path.join(PROF, f"{tag}_bench_{kind}.json")
    if os.path.exists(pmc) and os.path.exists(bench_file):
        line = json.loads(open(bench_file).read())
        line["roofline"]["traffic"] = json.load(open(pmc))["hbm_bytes_per_launch"]
        open(bench_file, "w").write(json.dumps(line) + "\n")


# SQ passes: matrix-core busy time and LDS behaviour of the dominant kernel (mean per dispatch)
def all_counters(kind, leg):
    f = find(f"pmc_{kind}_{leg}/**/*counter_collection.csv")
    if not f:
        return {}
    acc = {}
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"].split("(")[0]
            s = acc.setdefault(name, {}).setdefault(r["Counter_Name"], [0, 0.0])
            s[0] += 1
            s[1] += float(r["Counter_Value"])
    return {k: {c: v / n for c, (n, v) in d.items()} for k, d in acc.items()}


sq_rows = []
for kind in ("mel", "mfcc", "dct", "cqt", "stft"):
    bench_file = os.path.join(PROF, f"{tag}_bench_{kind}.json")
    if not os.path.exists(bench_file):
        continue
    kname = json.loads(open(bench_file).read())["roofline"]["kernel"]
    merged = {}
    for leg in ("SQ", "LDS"):
        tab = all_counters(kind, leg)
        kern = next((k for k in tab if kname in k), None)
        if kern:
            merged.update(tab[kern])
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
    if merged.get("SQ_WAVE_CYCLES") and merged.get("SQ_WAIT_INST_LDS") is not None:
        derived["wait_inst_lds_over_wave_cycles"] = merged["SQ_WAIT_INST_LDS"] / merged["SQ_WAVE_CYCLES"]
    for c, v in sorted(merged.items()):
        sq_rows.append(f'{kind},"{kname}",{c},{v:.1f}')
    for c, v in sorted(derived.items()):
        sq_rows.append(f'{kind},"{kname}",{c},{v:.5f}')
if sq_rows:
    with open(os.path.join(PROF, f"{tag}_sq_summary.csv"), "w") as fh:
        fh.write("kind,kernel,counter_or_ratio,mean_per_dispatch\n" + "\n".join(sq_rows) + "\n")
print("profiles/ refreshed for", tag)
