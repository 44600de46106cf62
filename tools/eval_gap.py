"""Kernel-level view of what runs between two consecutive objective kernels (from a rocprofv3 rocpd database)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if "k_objective32" in r[0] and "Lb0" in r[0] or ("k_objective32<" in r[0] and "false" in r[0])]
if len(idx) < 12:
    idx = [i for i, r in enumerate(rows) if "objective32" in r[0]]
a, b = idx[10], idx[11]
t0 = rows[a][1]
for name, st, en in rows[a:b + 1]:
    short = name.split("(")[0][-60:]
    print(f"{(st - t0) / 1e3:9.1f} us  +{(en - st) / 1e3:8.1f} us  {short}")
