"""Chronological phase summary of the LAST fit in a rocprofv3 rocpd database: consecutive
launches of the same kernel are merged; prints start offset, span, busy time and launch count."""
import re, sqlite3, sys
def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name); name = re.sub(r"^void ", "", name)
    return re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", name)
thresh = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 2e5     # optional: minimum busy time in us
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
# last fit starts at the last k_kernel_matrix pair (Lp matrix = small one)
idx = [i for i, r in enumerate(rows) if "k_row_sqnorms<" in r[0] or "k_row_sqnorms(" in r[0]]
starts = [i for i in idx]
begin = starts[-4] if len(starts) >= 4 else 0
rows = rows[begin:]
t0 = rows[0][1]
groups = []
for name, s, e in rows:
    n = short(name)
    if groups and groups[-1][0] == n:
        g = groups[-1]; g[2] = e; g[3] += e - s; g[4] += 1
    else:
        groups.append([n, s, e, e - s, 1])
print(f"{'kernel':40s} {'t_start_ms':>10s} {'span_ms':>9s} {'busy_ms':>9s} {'launches':>8s}")
for n, s, e, busy, cnt in groups:
    if busy > thresh or cnt > 20:
        print(f"{n[:40]:40s} {(s-t0)/1e6:10.2f} {(e-s)/1e6:9.2f} {busy/1e6:9.2f} {cnt:8d}")
print("total span ms", (rows[-1][2] - t0) / 1e6)
