#!/usr/bin/env python3
"""Idle time between the launches of a training step, from a rocprofv3 rocpd SQLite database.
A step = the dispatches from one `adamw_kernel` (exclusive) to the next (inclusive).  For every step with the modal number of
launches: span (first start -> last end), busy (union of the kernel intervals), idle = span - busy, launches.
usage: tools/rocpd_idle.py results.db  -> markdown on stdout"""
import sqlite3, statistics, subprocess, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = cur.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
steps, curstep = [], []
for st, en, name in rows:
    curstep.append((st, en, name))
    if "adamw_kernel" in name or "adam_kernel" in name:
        steps.append(curstep)
        curstep = []
if not steps:
    sys.exit("no optimiser launches in this trace")
def union_busy(s):
    busy, hi = 0, s[0][0]
    for st, en, _ in s:
        if en > hi:
            busy += en - max(st, hi)
            hi = en
    return busy


def dominant(s):
    tot = {}
    for st, en, name in s:
        tot[name] = tot.get(name, 0) + en - st
    return max(tot, key=tot.get)


groups = {}
for s in steps:
    groups.setdefault((len(s), dominant(s)), []).append(s)
print(f"{len(steps)} optimiser steps in the trace; groups of steps with the same number of launches (>= 3 steps):\n")
print("| launches per step | steps | dominant kernel | span ms | busy ms | idle ms | idle % | gaps > 1 us | largest gap us |")
print("|---|---|---|---|---|---|---|---|---|")
med = statistics.median
for (n, dom), ss in sorted(groups.items(), key=lambda kv: -len(kv[1])):
    if len(ss) < 3:
        continue
    ss = ss[1:]                       # the first step of a group follows a host-side pause (setup, synchronisation)
    span = [s[-1][1] - s[0][0] for s in ss]
    busy = [union_busy(s) for s in ss]
    gaps = [[b[0] - a[1] for a, b in zip(s, s[1:])] for s in ss]
    dom = subprocess.run(["c++filt", dom.replace(".kd", "")], capture_output=True, text=True).stdout.strip().replace("void ", "").split("(")[0][:60]
    idle = [a - b for a, b in zip(span, busy)]
    print(f"| {n} | {len(ss)} | {dom} | {med(span)/1e6:.3f} | {med(busy)/1e6:.3f} | {med(idle)/1e6:.3f} | {100*med(idle)/med(span):.1f} | "
          f"{med([sum(g > 1000 for g in gs) for gs in gaps]):.0f} | {med([max(gs) for gs in gaps])/1e3:.1f} |")
print("\n(the gaps are the HIP-event brackets that bench.py records around sampled launches inside its timed region; launches that "
      "follow each other directly start within 0.1 us of the end of their predecessor in this clock)")
