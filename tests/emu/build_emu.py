"""Build libmigan_emu.so (CPU fiber emulation of the product kernels) for the CPU test-suite.
Test infrastructure only: the package never loads this library."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "libmigan_emu.so")
SOURCES = [
    os.path.join(HERE, "migan_emu.cpp"),
    os.path.join(HERE, "hip_emu.h"),
    os.path.join(ROOT, "mi-gan_amd", "csrc", "migan_kernels.hpp"),
    os.path.join(ROOT, "mi-gan_amd", "csrc", "migan_host.hpp"),
    os.path.join(ROOT, "mi-gan_amd", "csrc", "migan_table.hpp"),
    os.path.join(ROOT, "mi-gan_amd", "csrc", "migan_pipeline.hpp"),
    os.path.join(ROOT, "mi-gan_amd", "csrc", "migan_pipe.hpp"),
    os.path.join(ROOT, "mi-gan_amd", "csrc", "migan_pipe_table.inc"),
    os.path.join(ROOT, "mi-gan_amd", "csrc", "migan_wide2.hpp"),
    os.path.join(ROOT, "mi-gan_amd", "csrc", "migan_wide2_table.inc"),
    os.path.join(ROOT, "mi-gan_amd", "csrc", "migan_k_slice.inc"),
    os.path.join(ROOT, "include", "migan_hip.h"),
    os.path.join(ROOT, "mi-gan_amd", "csrc", "comodgan_kernels.hpp"),
    os.path.join(ROOT, "mi-gan_amd", "csrc", "comodgan_host.hpp"),
    os.path.join(ROOT, "include", "comodgan_hip.h"),
]


def _compiler():
    for c in ("/opt/rocm/lib/llvm/bin/clang++",):
        if os.path.exists(c):
            return c
    return "clang++"


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(OUT):
        newest = max(os.path.getmtime(s) for s in SOURCES)
        if os.path.getmtime(OUT) >= newest:
            return OUT
    cmd = [_compiler(), "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-psabi",
           SOURCES[0], "-o", OUT, "-lpthread"]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
