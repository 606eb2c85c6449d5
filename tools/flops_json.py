"""rocprofv3 --pmc database -> profiles/flops_<config>.json (the MEASURED instruction mix bench.py's companion roofline reads).

    python tools/flops_json.py <results.db> <bench log of the same run> <config> [bench args] > flops_<config>.json

Per kernel (largest grid, mean over its launches in the run): wave-level counts of float64 ADD / MUL / FMA / TRANS
instructions, all VALU instructions, SQ_BUSY_CYCLES, SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES, and
    flop_per_launch = 64 lanes x (ADD + MUL + 2 FMA + TRANS)      (full EXEC mask assumed: an upper bound per instruction)
together with the number of genes a launch of that run worked on (the bench line's `genes_nonzero`), so that the
figure scales to a launch of another size.  The per-kernel fp64 issue bound on gfx950: a wave64 DP instruction
occupies its SIMD for 4 cycles (16 lanes per clock), FMA = 2 flop -> 256 CU x 4 SIMD x 16 x 2 x 2.4 GHz = 78.6 TFLOP/s.
"""
import json
import sqlite3
import sys

COUNTERS = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
            "SQ_INSTS_VALU_TRANS_F64", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"]


def short(name):
    return name.split("(")[0].replace("void ", "")


def main():
    db, log, cfg = sys.argv[1], sys.argv[2], sys.argv[3]
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, grid_size, counter_name, value from counters_collection").fetchall()
    agg = {}
    for n, g, c, v in rows:
        agg.setdefault(short(n), {}).setdefault(g, {}).setdefault(c, []).append(v)
    line = None
    for ln in open(log):
        if ln.startswith("{") and '"metric"' in ln:
            line = json.loads(ln)
    genes = None if line is None else line.get("genes_nonzero", line["config"]["genes_per_gpu"])
    kernels = {}
    for k, by_grid in agg.items():
        g = max(by_grid)  # the full-size launches
        c = {name: (sum(v) / len(v)) for name, v in by_grid[g].items()}
        if not all(x in c for x in COUNTERS[:5]):
            continue
        n_launch = len(by_grid[g][COUNTERS[0]])
        f64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_TRANS_F64"]
        flop = 64.0 * (c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + 2.0 * c["SQ_INSTS_VALU_FMA_F64"]
                       + c["SQ_INSTS_VALU_TRANS_F64"])
        kernels[k] = {"grid": int(g), "launches": n_launch, **{x: round(c.get(x, float("nan")), 1) for x in COUNTERS},
                      "f64_wave_instructions": round(f64, 1), "f64_share_of_valu": round(f64 / max(c["SQ_INSTS_VALU"], 1.0), 4),
                      "flop_per_launch": round(flop, 1)}
    out = {"config": cfg, "bench_args": sys.argv[4:], "genes_per_launch": genes,
           "source": "rocprofv3 --kernel-trace --pmc " + " ".join(COUNTERS) + " (tools/pmc_flops.sh; one step, one pass: 8 SQ slots)",
           "flop_definition": "64 x (ADD_F64 + MUL_F64 + 2 FMA_F64 + TRANS_F64) wave instructions, full EXEC assumed",
           "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["flop_per_launch"]))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
