#!/usr/bin/env python3
"""Register / LDS / scratch use of the kernels in a built object or library (developer tool).

    python tools/kernel_resources.py build/obj/dsq_k_irls.o [name-filter]

Finds the embedded gfx950 code objects (ELF images inside the fat binary), asks llvm-readelf for their
metadata notes and prints one line per kernel.
"""
import re, subprocess, sys, tempfile, os

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"

def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    blob = open(path, "rb").read()
    seen = set()
    for m in re.finditer(b"\x7fELF\x02\x01\x01", blob):
        off = m.start()
        if blob[off + 18:off + 20] != b"\xe0\x00":  # e_machine EM_AMDGPU
            continue
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(blob[off:])
            tmp = f.name
        try:
            out = subprocess.run([READELF, "--notes", tmp], capture_output=True, text=True).stdout
        finally:
            os.unlink(tmp)
        for blk in out.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
            name = g("name")
            if name in seen or flt not in name:
                continue
            seen.add(name)
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dem = re.sub(r"\(.*", "", dem)
            print(f"{dem:48s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>4s} sgpr_spill {g('sgpr_spill_count'):>3s} "
                  f"scratch {g('private_segment_fixed_size'):>5s} B  lds {g('group_segment_fixed_size'):>6s} B")

if __name__ == "__main__":
    main()
