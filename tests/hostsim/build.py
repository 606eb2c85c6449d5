"""Build tests/hostsim/_hostsim.so (g++, host only).  Test infrastructure, see hostsim.cpp."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(HERE, "hostsim.cpp")
OUT = os.path.join(HERE, "_hostsim.so")
INC = os.path.join(ROOT, "pydeseq2_amd", "csrc")


def build(force=False):
    deps = [SRC] + [os.path.join(INC, f) for f in os.listdir(INC) if f.endswith(".h")]
    if not force and os.path.exists(OUT) and all(
        os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps
    ):
        return OUT
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-I", INC, SRC,
           "-o", OUT]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
