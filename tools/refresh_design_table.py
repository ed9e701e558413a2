#!/usr/bin/env python3
"""Rewrites the measurement table and the counter paragraph of DESIGN.md section 6 from the files under profiles/.

    python tools/summarize_profiles.py r02 && python tools/refresh_design_table.py
"""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def ld(k):
    return json.loads(open(os.path.join(P, f"r02_bench_{k}.json")).read())


a = ld("all")
c = a["configs"]
ist, dc = ld("istft"), ld("dct")
sq = {}
for r in csv.reader(open(os.path.join(P, "r02_sq_summary.csv"))):
    if len(r) == 4 and r[2] in ("mfma_util", "lds_bank_conflict_over_active"):
        sq[(r[0], r[2])] = float(r[3])
ratio = {}
for k in ["stft", "istft", "mdct", "imdct", "mel", "mfcc", "cqt", "dct"]:
    p = json.load(open(os.path.join(P, f"pmc_{k}.json")))
    ratio[k] = p["hbm_bytes_per_launch"] / ld(k)["roofline"]["algorithmic_bytes_per_launch"]
tbl = f"""| kernel (config) | ms / launch r1 → r2 | Msamples/s | roofline |
|---|---|---|---|
| `k_stft_ft16` (config 2: 1024 × 10 s, reference layout) | 1.81–1.91 → {a['ms_per_step']:.2f} (1.50–1.70 over the boxes seen since the early request, 1.62–1.75 before) | {a['value']:.0f} | {a['roofline']['achieved']:.0f} GB/s algorithmic = {a['roofline']['frac']:.3f} of 8 TB/s ({a['roofline']['frac_of_achievable_6290']:.2f} of the 6.29 TB/s achievable; round 1's zero-compute kernel with this read / write pattern, timed over 20 launches: 1.68 ms) |
| `k_mel` (config 3) | 1.47–1.49 → {c['mel']['ms_per_step']:.2f} | {c['mel']['value']:.0f} | {c['mel']['roofline']['achieved']:.1f} TF (FFT 49.8 + issued MFMA 15.3 GFLOP) = {c['mel']['roofline']['frac']:.2f} of 157.3; HBM {c['mel']['roofline']['hbm']['frac']:.2f} |
| `k_mel`, mfcc (config 3) | 1.62–1.63 → {c['mfcc']['ms_per_step']:.2f} | {c['mfcc']['value']:.0f} | {c['mfcc']['roofline']['achieved']:.1f} TF = {c['mfcc']['roofline']['frac']:.2f}; HBM {c['mfcc']['roofline']['hbm']['frac']:.2f} |
| `k_mdct_ft32` (config 4) | 1.00–1.05 → {c['mdct']['ms_per_step']:.2f} | {c['mdct']['value']:.0f} | {c['mdct']['roofline']['achieved']:.0f} GB/s = {c['mdct']['roofline']['frac']:.3f} |
| `k_imdct` (config 4) | 1.12–1.13 → {c['imdct']['ms_per_step']:.2f} | {c['imdct']['value']:.0f} | {c['imdct']['roofline']['achieved']:.0f} GB/s = {c['imdct']['roofline']['frac']:.3f} (zero-compute kernel with this gather + stores: 0.72–0.89 ms depending on the box) |
| mdct + imdct round trip (config 4) | 2.12–2.18 → {c['mdct_imdct_roundtrip']['ms_per_step']:.2f} | {c['mdct_imdct_roundtrip']['value']:.0f} | residual 1.2e-06 (< 1e-5) |
| `k_cqt` (config 5 share: 1024 × 30 s per GPU) | 46.6 → {c['cqt']['ms_per_step']:.1f} | {c['cqt']['value']:.0f} | {c['cqt']['roofline']['achieved']:.1f} TF algorithmic = {c['cqt']['roofline']['frac']:.3f} of 157.3 (r1 0.26) |
| `k_istft_ft16` | 1.94–2.03 → {ist['ms_per_step']:.2f} | {ist['value']:.0f} | {ist['roofline']['achieved']:.0f} GB/s = {ist['roofline']['frac']:.3f} |
| `k_linear128` (`zaf.dct`, 16 384 × 1024) | 0.45 → {dc['ms_per_step']:.2f} | — | {dc['roofline']['achieved']:.0f} TF f32 MFMA = {dc['roofline']['frac']:.2f}; MFMA utilisation from the SQ counters {100 * sq.get(('dct', 'mfma_util'), 0):.1f} % |
"""
cb, ca, e2e = a["cpu_baseline"]["value"], a["cpu_baseline_all_cores"], a["end_to_end_pcie"]
txt = f"""CPU baselines in the same run (oracle ports of the per-clip NumPy path, GPU box host): stft {cb:.1f} Msamples/s on 1 core → {a['value'] / cb:.0f}×; all
usable cores ({ca['cores']} processes, BLAS threads 1): {ca['value']:.1f} Msamples/s → {a['value'] / ca['value']:.0f}×; per function under `configs.*.cpu_baseline` (mel {c['mel']['cpu_baseline']['value']:.1f}, mfcc {c['mfcc']['cpu_baseline']['value']:.1f}, mdct
{c['mdct']['cpu_baseline']['value']:.1f}, imdct {c['imdct']['cpu_baseline']['value']:.1f}, cqt {c['cqt']['cpu_baseline']['value']:.2f} Msamples/s on one core).  End to end over PCIe with page-locked buffers both ways: {e2e['value']:.0f} Msamples/s (128 clips,
H2D 57 GB/s, D2H 57 GB/s, serial on one stream) — two orders of magnitude below the device-resident figure, and never reported as `value`.
`bench.py` times 100 steps after 20 warm-up steps by default: with 20 / 3 the ~1 ms kernels were measured inside the clock ramp (K = 2 … 400
back-to-back launches of the MDCT: 0.81 / 0.87 / 0.85 / 0.80 / 0.78 ms per launch, `tools/b2b_test.py`); boxes differ by up to 10 % (the same binaries: stft 1.50–1.70 ms, mel 0.94–1.03 ms).
Counter evidence (`profiles/r02_pmc_summary.csv`, `r02_sq_summary.csv`): fabric traffic / algorithmic bytes stft {ratio['stft']:.2f}, istft {ratio['istft']:.2f}, mdct {ratio['mdct']:.2f},
imdct {ratio['imdct']:.2f} (rows 1728 B apart: the 128-B runs of the odd rows straddle two lines, both fetched whole, the other halves wanted one tile later when L2 has dropped them; reading those rows on the line grid would need 32 KB of stash per workgroup that neither the registers — 116 of 128 — nor LDS have, and the copy kernel with such reads is only 3–4 % faster, `tools/exp_imdctcopy.hip`: the re-fetched halves come from the memory-side cache), mel {ratio['mel']:.2f}, mfcc {ratio['mfcc']:.2f}, **cqt {ratio['cqt']:.2f} (round 1: 8.3)**, dct {ratio['dct']:.1f};
MFMA utilisation (busy cycles / kernel cycles / SIMDs) k_mel {100 * sq.get(('mel', 'mfma_util'), 0):.1f} %, mfcc {100 * sq.get(('mfcc', 'mfma_util'), 0):.1f} %, k_linear128 {100 * sq.get(('dct', 'mfma_util'), 0):.1f} %; LDS bank conflicts / active cycles
stft {sq.get(('stft', 'lds_bank_conflict_over_active'), 0):.3f}, mel {sq.get(('mel', 'lds_bank_conflict_over_active'), 0):.2f}, cqt {sq.get(('cqt', 'lds_bank_conflict_over_active'), 0):.2f}, k_linear {sq.get(('dct', 'lds_bank_conflict_over_active'), 0):.2f} (the LDS timing model behind these: `tools/exp_ldsbank.hip`,
`tools/lds_model.py`, profiles/r02_notes.md).

"""
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
a0 = s.index("| kernel (config) | ms / launch r1 → r2 | Msamples/s | roofline |")
a1 = s.index("CPU baselines in the same run")
s = s[:a0] + tbl + "\n" + s[a1:]
b0 = s.index("CPU baselines in the same run")
b1 = s.index("Round-1 text follows for the floors")
s = s[:b0] + txt + s[b1:]
open(path, "w").write(s)
print(tbl)
