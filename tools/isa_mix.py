#!/usr/bin/env python3
"""Static instruction mix of one kernel, per segment between s_barrier instructions.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -x hip -S --cuda-device-only -o k.s file.hip
    python tools/isa_mix.py k.s <mangled-name-prefix>
"""
import collections
import re
import sys

src, prefix = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and l.rstrip().split(":")[0].startswith(prefix) and ":" in l)
seg, segs = collections.Counter(), []
first = start
for i in range(start + 1, len(lines)):
    l = lines[i].strip()
    if not l or l.startswith(";") or l.startswith("."):
        if l.startswith(".LBB"):
            seg["label"] += 1
        continue
    op = l.split()[0]
    if op == "s_endpgm":
        segs.append((first, i, seg))
        break
    if op == "s_barrier":
        segs.append((first, i, seg))
        seg, first = collections.Counter(), i
        continue
    if op.startswith("v_mfma"):
        seg["mfma"] += 1
    elif op.startswith("v_pk_"):
        seg["valu_pk"] += 1
    elif op.startswith("v_"):
        seg["valu"] += 1
        seg["v:" + re.sub(r"_e32|_e64|_dpp|_sdwa", "", op)] += 1
    elif op.startswith("ds_"):
        seg["lds"] += 1
    elif op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        seg["vmem"] += 1
    elif op.startswith("s_waitcnt"):
        seg["waitcnt"] += 1
    elif op.startswith("s_"):
        seg["salu"] += 1
    else:
        seg["other"] += 1
for a, b, s in segs:
    main = {k: v for k, v in s.items() if not k.startswith("v:")}
    top = sorted(((v, k[2:]) for k, v in s.items() if k.startswith("v:")), reverse=True)[:8]
    print(f"lines {a - start:5d}-{b - start:5d}: " + " ".join(f"{k}={v}" for k, v in sorted(main.items())))
    if top and main.get("valu", 0) > 40:
        print("      " + " ".join(f"{k}={v}" for v, k in top))
