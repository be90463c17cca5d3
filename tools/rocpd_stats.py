#!/usr/bin/env python3
"""Kernel-time summary (the `--stats` view) from a rocprofv3 rocpd SQLite database.
usage: tools/rocpd_stats.py results.db [steps_in_run]  -> markdown table on stdout"""
import sqlite3, subprocess, sys
db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
                   f"max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(d.group_segment_size) "
                   f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
span = cur.execute(f"select max(end)-min(start) from {kd}").fetchone()[0]
print(f"total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches; first->last span {span/1e6:.3f} ms\n")
print("| kernel | calls | total ms | % | avg us | min us | max us | vgpr | agpr | lds |")
print("|---|---|---|---|---|---|---|---|---|---|")
for name, n, t, mn, mx, vg, ag, lds in rows:
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = dn.replace("void ", "").split("(")[0][:70]
    print(f"| {dn} | {n} | {t/1e6:.3f} | {100*t/tot:.1f} | {t/n/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {vg} | {ag} | {lds} |")
