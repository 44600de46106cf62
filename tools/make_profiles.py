"""Turn the rocprofv3 databases of tools/profile_round.sh into the committed evidence files:

    profiles/objective_traffic.json   HBM bytes per launch of the two objective kernels (FETCH_SIZE x 2 + WRITE_SIZE,
                                      the gfx950 correction of guides/MI355X_MICROARCH.md 'HBM')
    profiles/mfma_util.json           fp64 matrix-pipe utilisation of the MFMA kernels: v_mfma_f64_16x16x4 issues
                                      x 2048 flop / launch time, against the 78.6 TFLOP/s peak

    python tools/make_profiles.py <stats.db> <fetch.db> <write.db> <sq.db> <n_local> <m> <ldl> <tag>
"""
import json
import re
import sqlite3
import sys

FP64_PEAK_TF = 78.6
INT8_PEAK_TOPS = 3944.0


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", name)


def counters(path):
    con = sqlite3.connect(path)
    out = {}
    for k, c, v, d in con.execute("select kernel_name, counter_name, value, duration from counters_collection"):
        out.setdefault(short(k), {}).setdefault(c, []).append((v, d))
    return out


def stats(path):
    con = sqlite3.connect(path)
    out = {}
    for k, n, avg in con.execute("select name, count(*), avg(duration) from kernels group by name"):
        out[short(k)] = (n, avg)
    return out


def full_launches(vals, frac=0.5):
    """launches that really streamed the buffer (the solver's gated no-op launches last microseconds)"""
    dmax = max(d for _, d in vals)
    return [(v, d) for v, d in vals if d > frac * dmax]


def main(a):
    st, fe, wr, sq = stats(a[0]), counters(a[1]), counters(a[2]), counters(a[3])
    n_local, m, ldl, tag = int(a[4]), int(a[5]), int(a[6]), a[7]
    def traffic(prefix, bytes_per_elem):
        kf = [k for k in fe if k.startswith(prefix) and "FETCH_SIZE" in fe[k]]
        best = max(kf, key=lambda k: max(d for _, d in fe[k]["FETCH_SIZE"]))
        f = full_launches(fe[best]["FETCH_SIZE"])
        w = full_launches(wr[best]["WRITE_SIZE"])
        fkb = sum(v for v, _ in f) / len(f)
        wkb = sum(v for v, _ in w) / len(w)
        con = sqlite3.connect(a[0])
        rows = con.execute("select duration from kernels where name like ?", ("%" + best.split("<")[0] + "<" + best.split("<")[1].split(">")[0] + ">%",)).fetchall()
        durs = [r[0] for r in rows]
        dmax = max(durs)
        durs = [d for d in durs if d > 0.5 * dmax]
        return {"kernel": best, "fetch_size_kb_raw": fkb, "write_size_kb_raw": wkb,
                "hbm_bytes_per_launch": 2.0 * fkb * 1024.0 + wkb * 1024.0,
                "algorithmic_bytes_per_launch": n_local * ldl * bytes_per_elem,
                "avg_launch_us_rocprof": sum(durs) / len(durs) / 1e3, "launches_averaged": len(durs)}
    t64 = traffic("k_objective<", 8)
    t32 = traffic("k_objective32<", 4)
    out = {"n_local": n_local, "m": m, "ldl": ldl, **t64,
           "correction": "FETCH_SIZE x2 (gfx950: 128-B requests of a 16 B/lane coalesced stream are tallied at 64 B, "
                         "guides/MI355X_MICROARCH.md 'HBM'); WRITE_SIZE as reported",
           "source": f"profiles/{tag}_bench_c3_summary.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes "
                     "with --kernel-trace only, bench.py --steps 1 --warmup 0; avg launch from the --kernel-trace --stats pass; "
                     "launches the device-resident solver gated off are excluded)",
           "fp32_passes": t32}
    json.dump(out, open("profiles/objective_traffic.json", "w"), indent=1)
    # MFMA utilisation
    mf = {"peak_tflops_fp64": FP64_PEAK_TF, "peak_tops_int8": INT8_PEAK_TOPS,
          "definition": "SQ_INSTS_MFMA x flop per instruction / launch duration.  fp64 kernels: every matrix instruction is "
          "v_mfma_f64_16x16x4_f64 (2048 flop), frac = that / 78.6 TFLOP/s; on gfx950 the fp64 matrix pipe runs at the fp64 "
          "vector rate, so frac is also the share of the CU's fp64 issue slots spent in MFMAs.  k_gram_i8: every matrix "
          "instruction is v_mfma_i32_32x32x32_i8 (65536 integer operations), frac = that / 3944 TOP/s (the measured "
          "int8 ceiling of guides/MI355X_MICROARCH.md).  k_rowmin_f16x3 (1-NN pre-filter, k-means assignment): "
          "v_mfma_f32_32x32x16_f16 (32768 flop) against the 2500 TFLOP/s dense fp16 peak.", "kernels": {}}
    for k, cs in sq.items():
        if "SQ_INSTS_MFMA" not in cs:
            continue
        vals = cs["SQ_INSTS_MFMA"]
        if max(v for v, _ in vals) <= 0:
            continue
        v, d = max(vals, key=lambda q: q[1])            # the largest launch of this kernel
        tot_v, tot_d = sum(q[0] for q in vals), sum(q[1] for q in vals)
        i8 = k.startswith("k_gram_i8")
        f16 = "k_rowmin_f16x3" in k                      # v_mfma_f32_32x32x16_f16: 32768 flop, dense fp16 peak 2500 TFLOP/s
        per, peak = (65536, INT8_PEAK_TOPS) if i8 else ((32768, 2500.0) if f16 else (2048, FP64_PEAK_TF))
        mf["kernels"][k] = {"launches": len(vals), "largest_launch_us": d / 1e3, "largest_launch_tflops": v * per / d / 1e3,
                            "largest_launch_frac_of_peak": v * per / d / 1e3 / peak,
                            "all_launches_tflops": tot_v * per / tot_d / 1e3,
                            "all_launches_frac_of_peak": tot_v * per / tot_d / 1e3 / peak,
                            "unit": "TOP/s (int8)" if i8 else ("TFLOP/s (fp16, fp32 accumulate)" if f16 else "TFLOP/s (fp64)")}
    mf["source"] = f"rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace, bench.py --steps 1 --warmup 0 ({tag})"
    json.dump(mf, open("profiles/mfma_util.json", "w"), indent=1)
    print(json.dumps({"traffic64": t64["hbm_bytes_per_launch"] / t64["algorithmic_bytes_per_launch"],
                      "traffic32": t32["hbm_bytes_per_launch"] / t32["algorithmic_bytes_per_launch"],
                      "mfma": {k: round(v["largest_launch_frac_of_peak"], 3) for k, v in mf["kernels"].items()}}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
