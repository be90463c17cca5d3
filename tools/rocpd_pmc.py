#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database (separate --pmc pass).
usage: tools/rocpd_pmc.py results.db  -> {kernel: (calls, avg_value)}; prints a markdown table"""
import sqlite3, subprocess, sys


def per_kernel(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    pe = [t for t in tabs if "pmc_event" in t][0]
    ip = [t for t in tabs if "info_pmc" in t][0]
    rows = cur.execute(f"select s.kernel_name, i.name, count(*), avg(p.value), min(p.value), max(p.value) from {pe} p "
                       f"join {kd} d on p.event_id = d.event_id join {ks} s on d.kernel_id = s.id "
                       f"join {ip} i on p.pmc_id = i.id group by s.kernel_name, i.name order by 4 desc").fetchall()
    out = {}
    for name, cname, n, avg, mn, mx in rows:
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = dn.replace("void ", "").split("(")[0][:64]
        out[dn] = (cname, n, avg, mn, mx)
    return out


if __name__ == "__main__":
    res = per_kernel(sys.argv[1])
    print("| kernel | counter | calls | avg (KB) | min | max |\n|---|---|---|---|---|---|")
    for k, (c, n, a, mn, mx) in res.items():
        print(f"| {k} | {c} | {n} | {a:.1f} | {mn:.1f} | {mx:.1f} |")
