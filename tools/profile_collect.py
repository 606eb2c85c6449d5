"""Assemble profiles/<tag>_<config>.txt and profiles/traffic_<config>.json from gpurun_out/<tag>/
(written on the GPU box by tools/profile_round.sh).

    python tools/profile_collect.py <tag> [config] [extra bench logs ...]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    cfg = sys.argv[2] if len(sys.argv) > 2 else "c3"
    extra = sys.argv[3:]
    src = os.path.join(ROOT, "gpurun_out", tag)
    bench = [l for l in open(os.path.join(src, "bench_prof.log")) if l.startswith("{")][-1].strip()
    extra_args = os.environ.get("PROFILE_BENCH_ARGS", "")
    out = [f"# {tag} — rocprofv3 --kernel-trace --stats -- python bench.py --config {cfg} --steps 5 --warmup 2 --no-extras"
           f"{(' ' + extra_args) if extra_args else ''}  (MI355X)",
           "# bench line of the same (profiled) run:", bench, "", open(os.path.join(src, "stats.txt")).read(),
           "# separate PMC passes (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE; 1 step, no warmup).",
           "# Units KiB as reported; per MI355X_MICROARCH.md FETCH_SIZE is doubled on gfx950, WRITE_SIZE taken as is."]
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        out.append(open(os.path.join(src, f"pmc_{c}.txt")).read())
    for f in extra:
        lines = [l for l in open(f) if l.startswith("{")]
        if lines:
            out.append(f"# un-profiled bench line of the same build ({os.path.basename(f)}):")
            out.append(lines[-1].strip())
    path = os.path.join(ROOT, "profiles", f"{tag}_{cfg}.txt")
    open(path, "w").write("\n".join(out) + "\n")

    fetch = json.load(open(os.path.join(src, "pmc_FETCH_SIZE.json")))
    write = json.load(open(os.path.join(src, "pmc_WRITE_SIZE.json")))
    # the dispersion stage: the row kernel (four genes per wavefront) + the continuation of its parked fits where the
    # design takes them, else the full-size k_alpha launches (largest grid)
    rows = [k for k in fetch if k.startswith("dsq::k_alpha_rows<") or k.startswith("dsq::k_alpha_rows_c<")]
    mix = [k for k in fetch if k.startswith("dsq::k_alpha_mix<")]
    if mix:  # categorical + continuous designs: k_alpha_mix, the main launch + the continuation of its parked fits (round 5:
        # its own instantiation, one gene per workgroup) = one STAGE, which is what bench.py's full_launch_ms times
        parts = sorted(mix, key=lambda k: -fetch[k])[:2]
    elif rows:
        cont = [k for k in fetch if k.startswith("dsq::k_alpha_wg<")][:1]
        if not cont:  # the many-cell row kernel's parked fits are continued by k_alpha (largest launch of it)
            ka = [k for k in fetch if k.startswith("dsq::k_alpha<")]
            cont = [max(ka, key=lambda k: int(k.split("@")[1]))] if ka else []
        parts = [max(rows, key=lambda k: fetch[k])] + cont
    else:
        keys = [k for k in fetch if k.startswith("dsq::k_alpha<")]
        parts = [max(keys, key=lambda k: int(k.split("@")[1]))]
    f_kib = sum(fetch[k] for k in parts)
    w_kib = sum(write.get(k, 0.0) for k in parts)
    traffic = {
        "k_alpha_hbm_bytes_per_launch": int((2.0 * f_kib + w_kib) * 1024),
        "source": f"profiles/{tag}_{cfg}.txt: (2*FETCH_SIZE + WRITE_SIZE) KiB of the dispersion stage's kernels "
                  f"({' + '.join(k.split('@')[0] for k in parts)}; per STAGE = main launch + continuation, mean over the "
                  "genewise and the MAP stage; gfx950 "
                  "FETCH_SIZE x2 correction of MI355X_MICROARCH.md - calibrated there for 16-byte-per-lane streaming "
                  "reads; the row kernel reads 4 bytes per lane, so the absolute is uncalibrated)",
        "fetch_kib": f_kib, "write_kib": w_kib, "kernels": parts,
        "genes_per_launch": json.loads(bench)["config"].get("genes_per_gpu"),
    }
    json.dump(traffic, open(os.path.join(ROOT, "profiles", f"traffic_{cfg}.json"), "w"), indent=1)
    print(path, traffic)


if __name__ == "__main__":
    main()
