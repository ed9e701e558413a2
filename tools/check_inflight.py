#!/usr/bin/env python3
"""Check that the destination registers of asm-issued loads are not touched before the matching s_waitcnt.

    python tools/check_inflight.py kernel.s <first-load-line> <wait-line>
Lines are 1-based positions in the file; loads are the buffer_load/global_load lines at or after <first-load-line> up to the
first instruction that is not part of the request block.  Reports every instruction between the last load and <wait-line>
that names one of the destination VGPRs."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
lo, hi = int(sys.argv[2]) - 1, int(sys.argv[3]) - 1


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


dest, last = set(), lo
for i in range(lo, hi):
    t = lines[i].split()
    if t and re.match(r"(buffer|global)_load", t[0]):
        dest |= regs(t[1].rstrip(","))
        last = i
bad = 0
for i in range(last + 1, hi):
    l = lines[i].split(";")[0]
    t = l.split()
    if not t or t[0].startswith("."):
        continue
    used = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", l):
        used |= regs(tok)
    if used & dest:
        bad += 1
        print(f"{i + 1}: {l.strip()}   <- touches {sorted(used & dest)}")
print(f"{len(dest)} destination VGPRs, {bad} instructions touch them before the wait")
sys.exit(1 if bad else 0)
