"""The longest individual launches of the small-matrix phase in the LAST fp64 C3 step of a tools/one_step.py trace."""
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cols = [c[1] for c in con.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else "0")
gy = gx.replace("x", "y") if gx != "0" else "0"
wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else "1")
rows = con.execute(f"select name, start, end, {gx}, {gy}, {wx} from kernels order by start").fetchall()
print("columns:", cols)
km = [i for i, r in enumerate(rows) if "k_kernel_matrix_rows" in r[0] and r[2] - r[1] > 5e6]
rows = rows[km[-1] - 6:]
t0 = rows[0][1]
sel = [r for r in rows if "k_dgemm" in r[0] or "gram" in r[0]]
for r in sorted(sel, key=lambda r: r[1] - r[2])[:24]:
    n = re.sub(r"\(.*", "", r[0])
    print(f"{n:42s} at +{(r[1] - t0) / 1e6:8.2f} ms  {(r[2] - r[1]) / 1e3:9.1f} us  workgroups {r[3] // max(r[5], 1)} x {r[4]}")
