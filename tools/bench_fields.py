#!/usr/bin/env python3
"""Run bench.py with the given arguments and print a few fields of its JSON line (or the tail of its stderr when there is none).
usage: [ENV=..] bench_fields.py <bench args...>"""
import json
import subprocess
import sys
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + sys.argv[1:], capture_output=True, text=True)
lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
if not lines:
    print("NO JSON LINE (rc %d)\n%s\n%s" % (p.returncode, p.stdout[-1500:], p.stderr[-2500:]))
    sys.exit(1)
d = json.loads(lines[-1])
print("%s %s | value %.1f  sustained %s | settle %s | detail %s | ranks %s, %s" % (
    os.environ.get("OSVOS_DP_BACKEND", ""), " ".join(sys.argv[1:]), d["value"], (d.get("sustained") or {}).get("value"), d.get("setup_settle_steps"),
    d.get("timed_region_detail"), d["config"].get("rccl_ranks_seen"), d["config"].get("grad_allreduce")))
