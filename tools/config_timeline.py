"""Every launch of the LAST timed step of a `bench.py --config cN` kernel trace, in order (start offset, duration, gap to the
previous launch, workgroups, kernel), followed by the per-kernel totals of that step and the idle time between launches.
    python tools/config_timeline.py <rocprofv3 db> [max_lines]"""
import re, sqlite3, sys
from collections import defaultdict
con = sqlite3.connect(sys.argv[1])
cols = [c[1] for c in con.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else "0")
wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else "1")
rows = con.execute(f"select name, start, end, {gx}, {wx} from kernels order by start").fetchall()
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)[:72]
# a step begins with the big kernel-matrix launch (cov(x, xu)); steps = its launches within 2x of the longest
km = [i for i, r in enumerate(rows) if "k_kernel_matrix" in r[0]]
dmax = max(rows[i][2] - rows[i][1] for i in km)
big = [i for i in km if rows[i][2] - rows[i][1] > 0.5 * dmax]
# the last COMPLETE step: between the last two big launches (the very last one may be followed by untimed extras)
a, b = (big[-2], big[-1]) if len(big) >= 2 else (big[-1], len(rows))
step = rows[max(a - 6, 0):max(b - 6, 0)]
t0, prev = step[0][1], step[0][1]
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 400
tot, idle = defaultdict(lambda: [0, 0.0]), 0.0
for k, (n, s, e, g, w) in enumerate(step):
    if k < limit:
        print(f"+{(s - t0) / 1e6:9.3f} ms {(e - s) / 1e3:9.1f} us  gap {(s - prev) / 1e3:7.1f}  wg {g // max(w, 1):6d}  {short(n)}")
    idle += max(s - prev, 0) / 1e3
    tot[short(n)][0] += 1; tot[short(n)][1] += (e - s) / 1e3
    prev = e
span = (step[-1][2] - t0) / 1e3
print(f"\nstep span {span / 1e3:.3f} ms, {len(step)} launches, kernels busy {sum(v[1] for v in tot.values()) / 1e3:.3f} ms, idle between launches {idle / 1e3:.3f} ms")
for n, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:74s} {c:5d} {us / 1e3:9.3f} ms")
