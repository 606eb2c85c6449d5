"""Summarise rocprofv3 rocpd (sqlite) outputs into text/JSON kept under profiles/.

    python tools/rocprof_summary.py stats  <results.db>            -> per-kernel time table
    python tools/rocprof_summary.py pmc    <results.db> <counter>  -> per-kernel counter sums
"""
import json
import sqlite3
import sys


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n


def stats(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    print(f"{'kernel':48s} {'calls':>6s} {'total_us':>12s} {'avg_us':>12s} {'pct':>7s}")
    for n, c, t, a, p in rows:
        print(f"{short(n):48s} {c:6d} {t / 1e3 if t > 1e7 else t:12.1f} {a / 1e3 if t > 1e7 else a:12.1f} {p:7.2f}")
    # per-dispatch detail of the dominant kernel
    q = ("select kernel_name, grid_size, vgpr_count, accum_vgpr_count, sgpr_count, lds_block_size, scratch_size, "
         "(end-start) from kernels order by start")
    try:
        det = con.execute(q).fetchall()
        seen = {}
        for n, g, v, av, s, l, sc, d in det:
            k = short(n)
            seen.setdefault(k, []).append((g, v, av, s, l, sc, d))
        print("\nper-kernel resources (grid of the largest launch, VGPR, AGPR, SGPR, LDS, scratch) and largest-launch mean us")
        for k, v in seen.items():
            gmax = max(x[0] for x in v)
            big = [x for x in v if x[0] == gmax]
            print(f"{k:48s} grid {gmax:9d} vgpr {big[0][1]} agpr {big[0][2]} sgpr {big[0][3]} lds {big[0][4]} "
                  f"scratch {big[0][5]}  n_big {len(big)} mean_us {sum(x[6] for x in big) / len(big) / 1e3:.1f}")
    except Exception as e:  # schema differences
        print("detail query failed:", e)


def pmc(db, counter):
    con = sqlite3.connect(db)
    rows = con.execute(
        "select kernel_name, grid_size, value from counters_collection where counter_name=?", (counter,)).fetchall()
    agg = {}
    for n, g, v in rows:
        agg.setdefault((short(n), g), []).append(v)
    out = {}
    print(f"{'kernel':48s} {'grid':>10s} {'launches':>8s} {'mean ' + counter:>18s}")
    for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:48s} {g:10d} {len(v):8d} {sum(v) / len(v):18.1f}")
        out[f"{k}@{g}"] = sum(v) / len(v)
    return out


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        json.dump(pmc(sys.argv[2], sys.argv[3]), open(sys.argv[4], "w"), indent=1) if len(sys.argv) > 4 else pmc(sys.argv[2], sys.argv[3])
