"""Idle gaps between consecutive kernels of a rocprofv3 --kernel-trace run (rocpd sqlite):

    python tools/rocprof_gaps.py <results.db> [min_gap_us]

For every pair (previous kernel -> next kernel) whose start-to-end gap on the device timeline exceeds
min_gap_us (default 3): count, total and mean idle time.  Shows where host synchronisations / launch latency
leave the GPU empty inside a step.  Kernels of all streams are merged into one timeline (busy = union)."""
import sqlite3
import sys


def short(name):
    return name.split("(")[0].replace("void ", "")[:60]


def main():
    db = sys.argv[1]
    min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
    con = sqlite3.connect(db)
    cur = con.execute("select * from kernels limit 1")
    cols = [d[0] for d in cur.description]
    name_col = "name" if "name" in cols else ("kernel_name" if "kernel_name" in cols else None)
    if name_col is None or "start" not in cols or "end" not in cols:
        print("columns of `kernels`:", cols)
        return
    rows = con.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    print(f"{len(rows)} dispatches; columns: {cols}")
    agg = {}
    busy_end = rows[0][2]
    prev = rows[0][0]
    total_gap = 0.0
    for n, s, e in rows[1:]:
        gap = (s - busy_end) / 1e3
        if gap > min_gap and gap < 2000.0:  # longer pauses are between steps / phases of the bench
            k = (short(prev), short(n))
            a = agg.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += gap
            total_gap += gap
        if e > busy_end:
            busy_end = e
            prev = n
    print(f"idle gaps in ({min_gap}, 2000) us: total {total_gap / 1e3:.3f} ms")
    print(f"{'previous kernel -> next kernel':100s} {'n':>5s} {'total_us':>10s} {'mean_us':>8s}")
    for (a, b), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"{(a + ' -> ' + b):100s} {c:5d} {t:10.1f} {t / c:8.1f}")


if __name__ == "__main__":
    main()
