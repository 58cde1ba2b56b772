#!/usr/bin/env python3
"""Per-step kernel timeline out of a rocprofv3 kernel trace (rocpd sqlite): span, time with any kernel running, idle gaps, forward / backward
split and the launches of one step with their stream -- what the dispatch order and the cross-stream event hops cost.
usage: step_timeline.py <results.db> [step index, default 20] [from_us to_us ...]   (a step starts at its nchw_to_nhwc launch)"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
step = int(sys.argv[2]) if len(sys.argv) > 2 else 20
windows = [(float(sys.argv[i]), float(sys.argv[i + 1])) for i in range(3, len(sys.argv) - 1, 2)]
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
ks = [t for t in tabs if "kernel_symbol" in t.lower()][-1]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
namecol = "kernel_name" if "kernel_name" in cols else cols[-1]
disp = list(db.execute("select s.%s, d.start, d.end, d.stream_id from rocpd_kernel_dispatch d join %s s on d.kernel_id = s.id order by d.start" % (namecol, ks)))
starts = [i for i, d in enumerate(disp) if "nchw_to_nhwc" in d[0]]


def short(n):
    return re.sub(r"^\d+", "", n.replace("_ZN12_GLOBAL__N_1", "").replace("_ZN2at6native", "at::native::"))[:72]


def analyze(si, verbose):
    ev = disp[starts[si]:starts[si + 1]]
    t0, t1 = ev[0][1], max(e[2] for e in ev)
    iv = sorted((e[1], e[2]) for e in ev)
    busy, (cs, ce), gaps = 0, iv[0], []
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            gaps.append((s - ce, ce - t0))
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    tl = next(e[1] for e in ev if "cbce_count" in e[0])
    period = (disp[starts[si + 1]][1] - t0) / 1e3
    print("step %d: period (start to the next step's start) %.1f us, span %.1f us, a kernel running %.1f us, %d idle gaps totalling %.1f us; forward (until the loss kernels) %.1f us, rest %.1f us"
          % (si, period, (t1 - t0) / 1e3, busy / 1e3, len(gaps), sum(g[0] for g in gaps) / 1e3, (tl - t0) / 1e3, (t1 - tl) / 1e3))
    print("   largest gaps (us @ offset us): %s" % [(round(g[0] / 1e3, 1), round(g[1] / 1e3)) for g in sorted(gaps, reverse=True)[:8]])
    if verbose:
        print("   %8s %8s  stream  kernel" % ("start", "us"))
        for e in ev:
            off = (e[1] - t0) / 1e3
            if not windows or any(a <= off <= b for a, b in windows):
                print("   %8.1f %8.1f  s%-5d %s" % (off, (e[2] - e[1]) / 1e3, e[3], short(e[0])))


per = sorted((disp[starts[i + 1]][1] - disp[starts[i]][1]) / 1e3 for i in range(max(0, step - 40), min(len(starts) - 1, step + 40)))
print("step periods around step %d (%d steps): median %.1f us, min %.1f, p90 %.1f, max %.1f" % (step, len(per), per[len(per) // 2], per[0], per[int(len(per) * 0.9)], per[-1]))
for si in (step - 2, step + 2):
    analyze(si, False)
analyze(step, True)
