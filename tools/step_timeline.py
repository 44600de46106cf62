"""Every launch of the LAST fp64 C3 step of a tools/one_step.py kernel trace, in order: start offset, duration, gap to the
previous launch, workgroups, kernel.   python tools/step_timeline.py <db> [first_ms last_ms]"""
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cols = [c[1] for c in con.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else "0")
gy = gx.replace("x", "y") if gx != "0" else "0"
wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else "1")
rows = con.execute(f"select name, start, end, {gx}, {gy}, {wx} from kernels order by start").fetchall()
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)[:64]
km = [i for i, r in enumerate(rows) if "k_kernel_matrix_rows" in r[0] and r[2] - r[1] > 5e6]
# the step begins with cov(xu, xu) -- the previous k_kernel_matrix_rows launch (small) -- and its row norms
kuu = [i for i, r in enumerate(rows[:km[-1]]) if "k_kernel_matrix_rows" in r[0]]
rows = rows[(kuu[-1] - 3) if kuu else (km[-1] - 6):]
t0 = rows[0][1]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
prev = rows[0][1]
for n, s, e, g0, g1, w in rows:
    at = (s - t0) / 1e6
    if lo <= at <= hi:
        print(f"+{at:9.3f} ms {(e - s) / 1e3:9.1f} us  gap {(s - prev) / 1e3:7.1f}  wg {g0 // max(w, 1):6d} x {g1:<3d} {short(n)}")
    prev = e
