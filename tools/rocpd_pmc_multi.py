#!/usr/bin/env python3
"""Per-kernel averages of ALL counters in a rocprofv3 rocpd database.  usage: tools/rocpd_pmc_multi.py results.db [filter]"""
import collections, sqlite3, subprocess, sys
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
pe = [t for t in tabs if "pmc_event" in t][0]
ip = [t for t in tabs if "info_pmc" in t][0]
rows = cur.execute(f"select s.kernel_name, i.name, count(*), avg(p.value), avg(d.end - d.start) from {pe} p "
                   f"join {kd} d on p.event_id = d.event_id join {ks} s on d.kernel_id = s.id "
                   f"join {ip} i on p.pmc_id = i.id group by s.kernel_name, i.name").fetchall()
tab = collections.defaultdict(dict)
dur = {}
for name, cname, n, avg, d in rows:
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("void ", "").split("(")[0][:60]
    if flt and flt not in dn:
        continue
    tab[dn][cname] = avg
    dur[dn] = d / 1e3
cols = sorted({c for v in tab.values() for c in v})
print("| kernel | us | " + " | ".join(cols) + " |")
print("|---|---|" + "---|" * len(cols))
for k, v in tab.items():
    print(f"| {k} | {dur[k]:.1f} | " + " | ".join(f"{v.get(c, float('nan')):.3g}" for c in cols) + " |")
