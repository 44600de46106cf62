"""Per-kernel averages of the counters in one or more rocprofv3 --pmc databases (one row per kernel and counter).

    python tools/pmc_summary.py <db> [<db> ...] [--match substr] [--min-us T] > profiles/rNN_pmc.txt

--min-us T: only launches that ran at least T microseconds (a kernel launched at two very different sizes).
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", name)


def main(argv):
    match = None
    if "--match" in argv:
        i = argv.index("--match")
        match = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    min_ns = 0.0
    if "--min-us" in argv:
        i = argv.index("--min-us")
        min_ns = 1e3 * float(argv[i + 1])
        argv = argv[:i] + argv[i + 2:]
    table = {}
    for path in argv:
        con = sqlite3.connect(path)
        rows = con.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) "
                           "from counters_collection where duration >= ? group by kernel_name, counter_name", (min_ns,)).fetchall()
        for k, c, n, v, d in rows:
            k = short(k)
            if match and match not in k:
                continue
            table.setdefault(k, {})[c] = (n, v, d)
    for k in sorted(table, key=lambda q: -max(v[2] * v[0] for v in table[q].values())):
        cs = table[k]
        n, _, d = next(iter(cs.values()))
        print(f"{k[:70]:70s} launches {n:5d}  avg {d / 1e3:10.1f} us")
        for c in sorted(cs):
            print(f"    {c:34s} {cs[c][1]:18.1f}")


if __name__ == "__main__":
    main(sys.argv[1:])
