"""Idle gaps of the GPU inside one fp64 C3 step: from a rocprofv3 kernel trace of tools/one_step.py (the LAST step of
the run), the largest gaps between consecutive kernels and the kernels on either side."""
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)[:46]
# last step: from the last big kernel-matrix launch back to its row norms
km = [i for i, r in enumerate(rows) if "k_kernel_matrix_rows" in r[0] and r[2] - r[1] > 5e6]
begin = km[-1] - 6
rows = rows[begin:]
t0, t1 = rows[0][1], rows[-1][2]
busy = sum(e - s for _, s, e in rows)
print(f"step span {1e-6 * (t1 - t0):.2f} ms, kernels busy {1e-6 * busy:.2f} ms, {len(rows)} launches, idle {1e-6 * (t1 - t0 - busy):.2f} ms")
gaps = sorted(((rows[i + 1][1] - rows[i][2], i) for i in range(len(rows) - 1)), reverse=True)[:25]
for g, i in gaps:
    print(f"{1e-3 * g:9.1f} us at +{1e-6 * (rows[i][2] - t0):8.2f} ms  after {short(rows[i][0])}  before {short(rows[i + 1][0])}")
small = sum(g for g, i in ((rows[i + 1][1] - rows[i][2], i) for i in range(len(rows) - 1)) if g < 20e3)
print(f"gaps under 20 us: {1e-6 * small:.2f} ms in total")
