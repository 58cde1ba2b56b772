#!/usr/bin/env python3
"""Instruction mix of ONE kernel of a gfx950 .s file (hipcc -S --cuda-device-only), split at labels and barriers.
usage: asm_mix.py <file.s> <kernel-name substring> [min instructions per segment]"""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 30
lines = open(path).read().split("\n")
s = next(i for i, l in enumerate(lines) if pat in l and re.match(r"^[_A-Za-z0-9]+:", l))
e = next(i for i in range(s, len(lines)) if "s_endpgm" in lines[i])


def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_cmp", "v_cndmask")): return "valu_cmp"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_store", "global_store")): return "vst"
    if op.startswith(("buffer_", "global_", "flat_")): return "vld"
    return "other"


seg, cur, name = [], collections.Counter(), "entry"
for l in lines[s + 1:e]:
    t = l.strip()
    if not t or t.startswith((";", "//")):
        continue
    if t.endswith(":") or re.match(r"^\.LBB\S+:", t):
        seg.append((name, cur)); cur = collections.Counter(); name = t.split(":")[0]
        continue
    if t.startswith("."):
        continue
    c = cls(t.split()[0])
    cur[c] += 1
    if c == "barrier":
        seg.append((name + "..barrier", cur)); cur = collections.Counter(); name = "after-barrier"
seg.append((name, cur))
tot = collections.Counter()
for name, c in seg:
    tot.update(c)
    if sum(c.values()) >= minn:
        print("%-28s %5d  %s" % (name, sum(c.values()), dict(sorted(c.items()))))
print("TOTAL", sum(tot.values()), dict(sorted(tot.items())))
