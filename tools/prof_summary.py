#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace into the text summary kept under profiles/.
usage: prof_summary.py <results.db> <steps> [out.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2])
rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
out = ["rocprofv3 --kernel-trace --stats  (command: python bench.py --steps 20 --warmup 5 --no-cpu-baseline; %d profiled steps incl. warm-up)" % steps,
       "%-96s %7s %12s %10s %6s" % ("kernel", "calls", "us/step", "avg_us", "%")]
for n, c, tot, avg, pct in rows:
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    out.append("%-96s %7d %12.1f %10.1f %6.2f" % (n[:96], c, tot / steps, avg, pct))
out.append("total kernel time per step: %.1f us" % (sum(r[2] for r in rows) / steps))
text = "\n".join(out)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(text + "\n")
print(text)
