"""Summarise a rocprofv3 rocpd database (bench_results.db) as text: per-kernel launch statistics
(`--kernel-trace --stats` runs) and per-kernel PMC counter averages (`--pmc` runs).

    python tools/rocpd_summary.py <db> [<db> ...] > profiles/rNN_summary.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", name)


def main(paths):
    for path in paths:
        con = sqlite3.connect(path)
        print(f"== {path}")
        rows = con.execute(
            "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
            "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
            "from kernels group by name order by sum(duration) desc").fetchall()
        total = sum(r[2] for r in rows) or 1.0
        print(f"{'kernel':58s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} "
              f"{'%':>6s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>6s} {'scr':>4s}")
        for r in rows:
            print(f"{short(r[0])[:58]:58s} {r[1]:6d} {r[2]/1e6:10.3f} {r[3]/1e3:10.2f} {r[4]/1e3:10.2f} "
                  f"{r[5]/1e3:10.2f} {100*r[2]/total:6.2f} {r[6]:5d} {r[7]:5d} {r[8]:5d} {r[9]:6d} {r[10]:4d}")
        try:
            pmc = con.execute(
                "select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
                "from counters_collection group by kernel_name, counter_name order by sum(value) desc").fetchall()
        except sqlite3.Error:
            pmc = []
        if pmc:
            print(f"\n{'kernel':58s} {'counter':>12s} {'n':>6s} {'avg':>16s} {'min':>16s} {'max':>16s} {'avg_us':>10s}")
            for r in pmc:
                print(f"{short(r[0])[:58]:58s} {r[1]:>12s} {r[2]:6d} {r[3]:16.2f} {r[4]:16.2f} {r[5]:16.2f} {r[6]/1e3:10.2f}")
        print()


if __name__ == "__main__":
    main(sys.argv[1:])
