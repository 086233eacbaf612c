"""Build libbpe_hip.so for gfx950 in-tree:  python -m minbpe_amd.build

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the
working tree (gpurun snapshots it)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", f) for f in ("bpe_api.hip", "synth.cpp", "split.cpp", "dedup.cpp", "utf8.cpp")]
DEPS = SRC + [os.path.join(HERE, "csrc", f) for f in ("bpe_kernels.hip", "bpe_device.h", "unicode_tables.h")] + [
    os.path.join(HERE, "csrc", sub, f) for sub in ("kernels", "api")
    for f in sorted(os.listdir(os.path.join(HERE, "csrc", sub)))] + [
    os.path.join(ROOT, "include", "bpe_hip.h")]
OUT = os.path.join(HERE, "lib", "libbpe_hip.so")


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and os.path.exists(OUT) and all(
            os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-pthread",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc"),
           *SRC, "-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
