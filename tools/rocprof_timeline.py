"""Timeline of one bench step from a rocprofv3 --kernel-trace run (rocpd sqlite):

    python tools/rocprof_timeline.py <results.db> [step_index]

Finds the k_logmeans dispatches (one per deseq2() pass), takes the pass number `step_index` (default: the last
but one) and lists every dispatch of it: offset from the step's start, duration, idle time since the previous
dispatch ended (all streams merged), kernel name."""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cols = [d[0] for d in con.execute("select * from kernels limit 1").description]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = con.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    starts = [i for i, r in enumerate(rows) if "k_logmeans" in r[0] and "pos" not in r[0]]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 2
    lo, hi = starts[k], starts[k + 1]
    t0 = rows[lo][1]
    busy_end = rows[lo][1]
    busy = 0.0
    print(f"step {k}: {hi - lo} dispatches, {(rows[hi][1] - t0) / 1e3:.1f} us from its first kernel to the next step's")
    print(f"{'offset_us':>10s} {'dur_us':>9s} {'idle_us':>8s}  kernel")
    for n, s, e in rows[lo:hi]:
        gap = (s - busy_end) / 1e3
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {max(gap, 0.0):8.1f}  {n.split('(')[0].replace('void ', '')[:70]}")
        if e > busy_end:
            busy += (e - max(s, busy_end)) / 1e3
            busy_end = e
    print(f"busy {busy:.1f} us; idle to the next step's first kernel {(rows[hi][1] - busy_end) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
