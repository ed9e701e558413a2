#!/usr/bin/env python3
"""Rewrites the measurement table of DESIGN.md section 6 (between the table:begin / table:end markers) from profiles/<tag>_bench_all*.json.

    python tools/summarize_profiles.py r04 && python tools/design_table.py r04
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
a = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_all.json")))
plain = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_all_plain.json")))   # carries the in-run traffic ratios
c, cp = a["configs"], plain["configs"]


def tr(k):
    r = cp[k]["roofline"].get("traffic_over_algorithmic")
    return "—" if r is None else f"{r:.3f}"


def fr(k):
    r = c[k]["roofline"]
    s = f"{r['frac']:.3f} {r['bound']}"
    if "mfma_frac" in r:   # HBM is the roof; how busy the two arithmetic pipes are stands beside it
        s += f" (mfma {r['mfma_frac']:.3f}, valu {r['valu_frac']:.3f})"
    if "valu_frac_real_input" in r:   # config 5: the vector pipe, by the reference's complex flops and by the real-input form's
        s += f" ({r['valu_frac_real_input']:.3f} on the real-input flops; {r['hbm_frac']:.3f} hbm)"
    return s


ta = plain["roofline"]["traffic"] / (a["roofline"]["achieved"] * 1e9 * a["roofline"]["kernel_ms"] * 1e-3) if a["roofline"].get("traffic") else None
rows = [
    "| config (driver line, " + tag + " box) | ms / step | value (Msamples/s) | roofline frac | traffic | cpu_baseline (1 core) |",
    "|---|---|---|---|---|---|",
    f"| 2 `stft` (headline) | {a['ms_per_step']:.4f} | {a['value']:,.0f} | {a['roofline']['frac']:.3f} hbm | {ta:.3f} | {a['cpu_baseline']['value']:.1f} |",
]
for label, k in ((l, q) for l, q in (("2 `istft`", "istft"), ("3 `mel`", "mel"), ("3 `mfcc`", "mfcc"), ("3 `mel_mfcc` (both outputs, one pass)", "mel_mfcc"), ("4 `mdct`", "mdct"), ("4 `imdct`", "imdct"), ("5 `cqt` (1024 × 30 s = one GPU's share)", "cqt")) if q in c):
    rows.append(f"| {label} | {c[k]['ms_per_step']:.3f} | {c[k]['value']:,.0f} | {fr(k)} | {tr(k)} | {c[k]['cpu_baseline']['value']:.1f} |")
rt = c["mdct_imdct_roundtrip"]
rows.append(f"| 4 mdct + imdct round trip | {rt['ms_per_step']:.3f} | {rt['value']:,.0f} | residual {rt['residual_max_abs']:.2e} | | |")
tbl = "\n".join(rows).replace(",", " ")
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
i, j = s.index("<!-- table:begin -->"), s.index("<!-- table:end -->")
open(path, "w").write(s[:i] + "<!-- table:begin -->\n" + tbl + "\n" + s[j:])
print(tbl)
