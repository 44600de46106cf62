"""Per-kernel totals of the LAST fp64 C3 step in a rocprofv3 kernel trace of tools/one_step.py."""
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)[:60]
km = [i for i, r in enumerate(rows) if "k_kernel_matrix_rows" in r[0] and r[2] - r[1] > 5e6]
# the step begins with cov(xu, xu) -- the previous k_kernel_matrix_rows launch (small) -- and its row norms
kuu = [i for i, r in enumerate(rows[:km[-1]]) if "k_kernel_matrix_rows" in r[0]]
rows = rows[(kuu[-1] - 3) if kuu else (km[-1] - 6):]
tot = {}
for n, s, e in rows:
    k = short(n); t = tot.setdefault(k, [0, 0.0]); t[0] += 1; t[1] += (e - s) / 1e6
print(f"step: {(rows[-1][2] - rows[0][1]) / 1e6:.2f} ms, kernels {sum(v[1] for v in tot.values()):.2f} ms")
for k, (c, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:32]:
    print(f"{k:60s} {c:5d} {ms:9.3f} ms")
