#!/usr/bin/env python3
"""Static instruction totals of every kernel in a .s file (hipcc -S --cuda-device-only): packed f32 ops, other VALU,
v_mov, scalar f32 add/sub/mul (packed candidates the compiler split), permlane swaps, transcendental ops, LDS, VMEM,
VGPRs and spills.  For A/B of a change to the shared FFT core over all kernels:

    python tools/isa_totals.py before.s after.s [name-filter]
"""
import collections
import re
import sys


def totals(path):
    out, cur, c = {}, None, None
    for line in open(path):
        if line.startswith("_Z") and ":" in line and not line.startswith("_ZZ"):
            cur = line.split(":")[0]
            c = out.setdefault(cur, collections.Counter())
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            c["_done"] = 1
        s = line.strip()
        m = re.match(r"; (NumVgprs|ScratchSize|SGPRSpill|NumSgprs): (\d+)", s) or re.match(r"\.(vgpr_count|sgpr_spill_count):\s+(\d+)", s)
        if m:
            c[m.group(1)] = int(m.group(2))
            continue
        if not s or s[0] in ";." or s.endswith(":") or c.get("_done"):
            continue
        op = s.split()[0]
        if op.startswith("v_pk_"):
            c["pk"] += 1
        elif op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            if op.startswith("v_mov_b"):
                c["mov"] += 1
            elif re.match(r"v_(add|sub|mul|fma|fmac|subrev)_f32", op):
                c["sf32"] += 1
            elif op.startswith("v_permlane"):
                c["perm"] += 1
            elif re.match(r"v_(sqrt|rsq|rcp|log|exp|sin|cos)_", op):
                c["trans"] += 1
            elif "64" in op:
                c["v64"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("buffer_", "global_", "flat_", "scratch_")):
            c["vmem"] += 1
        elif op == "s_nop":
            c["nop"] += 1
    return out


def short(name):
    m = re.match(r"_ZN4zafx\d+(k_[a-z0-9_]+?)I?(L.*)?E?v?P", name)
    d = re.sub(r"^_ZN4zafx\d+", "", name)
    return d[:44]


KEYS = ("pk", "valu", "mov", "sf32", "perm", "trans", "v64", "lds", "vmem", "nop", "NumVgprs", "ScratchSize")
if __name__ == "__main__":
    a = totals(sys.argv[1])
    b = totals(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].endswith(".s") else None
    flt = sys.argv[-1] if not sys.argv[-1].endswith(".s") else ""
    print(f"{'kernel':44s} " + " ".join(f"{k[:8]:>9s}" for k in KEYS))
    for name, c in a.items():
        if flt not in name:
            continue
        row = []
        for k in KEYS:
            if b is not None and name in b:
                row.append(f"{c[k]:4d}>{b[name][k]:<4d}")
            else:
                row.append(f"{c[k]:9d}")
        print(f"{short(name):44s} " + " ".join(row))
