"""Ahead-of-time build of libmigan_hip.so for gfx950 (MI355X) with hipcc.

In-tree on purpose: the .so sits next to its sources (mi-gan_amd/csrc/) so it travels with the
repository snapshot to the GPU box and shows up as a loaded in-tree native library.

The library is several translation units compiled in parallel: the host plan + C ABI + the small
kernels (migan_hip.hip) and one unit per slice of the fused-SeparableConv2d kernel table
(migan_k_slice.hip compiled with -DMIGAN_SLICE_G=<GEMM variant> -DMIGAN_SLICE_S=<storage format>).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from typing import List, Sequence, Tuple

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
OUT = os.path.join(CSRC, "libmigan_hip.so")
OBJ = os.path.join(CSRC, "_obj")
SOURCES = [os.path.join(CSRC, f) for f in ("migan_hip.hip", "migan_k_slice.hip", "migan_k_slice.inc", "migan_table.hpp",
                                            "migan_pipe.hip", "migan_pipe.hpp", "migan_pipe_table.inc",
                                            "migan_wide2.hip", "migan_wide2.hpp", "migan_wide2_table.inc",
                                            "migan_kernels.hpp", "migan_host.hpp", "migan_rt_hip.h",
                                            "comodgan_kernels.hpp", "comodgan_host.hpp", "migan_pipeline.hpp")] + [
    os.path.join(ROOT, "include", "migan_hip.h"), os.path.join(ROOT, "include", "comodgan_hip.h")]
ARCH = "gfx950"
# (GEMM variant, activation storage format) slices of the sepconv_kernel table; 16-bit storage is built for the fp16 GEMM variants (f16x2 = 2, f16 = 3)
SLICES: Sequence[Tuple[int, int]] = ((0, 0), (1, 0), (2, 0), (2, 1), (2, 2), (3, 1), (3, 2))
# -fno-honor-nans is NOT used: what happens to a non-finite value is what the source says (clamp4 / clamp1, migan_kernels.hpp: a NaN
# activation stays a NaN like Tensor.clamp in the reference module; -DMIGAN_NAN_CLAMP: v_med3_f32 turns it into -256 like the reference's
# CUDA plugin bias_act.cu:139), never whatever the optimiser derives from "NaNs cannot occur"; pinned by tests/test_gpu_robust.py
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC"] + os.environ.get("MIGAN_HIPCC_FLAGS", "").split()


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 for --offload-arch=gfx950)")


def units(extra: Sequence[str] = ()) -> List[Tuple[str, List[str]]]:
    """(object file, command) per translation unit"""
    cc = hipcc()
    out = [(os.path.join(OBJ, "migan_hip.o"), [cc, *FLAGS, *extra, "-c", os.path.join(CSRC, "migan_hip.hip")]),
           (os.path.join(OBJ, "migan_pipe.o"), [cc, *FLAGS, *extra, "-c", os.path.join(CSRC, "migan_pipe.hip")]),
           (os.path.join(OBJ, "migan_wide2.o"), [cc, *FLAGS, *extra, "-c", os.path.join(CSRC, "migan_wide2.hip")])]
    for g, s in SLICES:
        out.append((os.path.join(OBJ, f"migan_k_g{g}s{s}.o"),
                    [cc, *FLAGS, *extra, f"-DMIGAN_SLICE_G={g}", f"-DMIGAN_SLICE_S={s}", "-c", os.path.join(CSRC, "migan_k_slice.hip")]))
    return [(o, cmd + ["-o", o]) for o, cmd in out]


STAMP = OUT + ".flags"


def flags_digest(extra: Sequence[str] = ()) -> str:
    """what the library was compiled with (a measurement build with -DMIGAN_PHASE_PROF must never pass for the product)"""
    return hashlib.sha256(" ".join([*FLAGS, *extra, *(f"{g}.{s}" for g, s in SLICES)]).encode()).hexdigest()[:16]


def is_fresh(extra: Sequence[str] = ()) -> bool:
    if not os.path.exists(OUT) or not os.path.exists(STAMP):
        return False
    if open(STAMP).read().strip() != flags_digest(extra):
        return False
    return os.path.getmtime(OUT) >= max(os.path.getmtime(s) for s in SOURCES)


def build(force: bool = False, verbose: bool = False, extra: Sequence[str] = (), jobs: int = 0, lint: bool = True) -> str:
    if not force and is_fresh(extra):
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    us = units(extra)
    if os.path.exists(STAMP):
        os.remove(STAMP)                      # a failed build must not leave a stamp that vouches for a stale library

    def run(u):
        if verbose:
            print(" ".join(u[1]), flush=True)
        r = subprocess.run(u[1], cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if verbose and r.stderr:
            print(r.stderr, flush=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed ({r.returncode}) on {os.path.basename(u[0])}:\n{r.stderr[-4000:]}")
        return u[0]

    with ThreadPoolExecutor(max_workers=jobs or min(len(us), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(run, us))
    link = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", OUT]
    if verbose:
        print(" ".join(link), flush=True)
    r = subprocess.run(link, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed ({r.returncode}):\n{r.stderr[-4000:]}")
    if lint:
        # the packed-fp32 op_sel hazard cannot be expressed in the source: refuse a library that contains it
        import importlib.util
        spec = importlib.util.spec_from_file_location("migan_isa_lint", os.path.join(HERE, "isa_lint.py"))
        isa_lint = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(isa_lint)
        if isa_lint.available():
            try:
                isa_lint.check(OUT)
            except RuntimeError:
                os.replace(OUT, OUT + ".rejected")
                raise
    with open(STAMP, "w") as f:
        f.write(flags_digest(extra) + "\n")
    return OUT


NAN_CLAMP_OUT = os.path.join(CSRC, "libmigan_hip_nanclamp.so")


def _lint_or_reject(path: str) -> None:
    """the packed-fp32 op_sel hazard cannot be expressed in the source: a library that contains it is moved aside, never stamped"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("migan_isa_lint", os.path.join(HERE, "isa_lint.py"))
    isa_lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(isa_lint)
    if isa_lint.available():
        try:
            isa_lint.check(path)
        except RuntimeError:
            os.replace(path, path + ".rejected")
            raise


def build_variant(name: str, extra: Sequence[str], force: bool = False, verbose: bool = False, lint: bool = False) -> str:
    """another build of the same library under libmigan_hip_<name>.so (objects in csrc/_obj_<name>/): the opt-in NaN -> -256 variant
    (-DMIGAN_NAN_CLAMP) and the measurement builds (scripts/build_variant.py)"""
    out = os.path.join(CSRC, f"libmigan_hip_{name}.so")
    stamp = out + ".flags"
    if (not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == flags_digest(extra)
            and os.path.getmtime(out) >= max(os.path.getmtime(s) for s in SOURCES)):
        return out
    obj = os.path.join(CSRC, f"_obj_{name}")
    os.makedirs(obj, exist_ok=True)
    us = []
    for o, cmd in units(extra):
        o2 = os.path.join(obj, os.path.basename(o))
        us.append((o2, cmd[:-1] + [o2]))

    def run(u):
        if verbose:
            print(" ".join(u[1]), flush=True)
        r = subprocess.run(u[1], cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed ({r.returncode}) on {os.path.basename(u[0])}:\n{r.stderr[-4000:]}")
        return u[0]

    with ThreadPoolExecutor(max_workers=min(len(us), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(run, us))
    r = subprocess.run([hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", out], cwd=CSRC, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed ({r.returncode}):\n{r.stderr[-4000:]}")
    if os.path.exists(stamp):
        os.remove(stamp)
    if lint:
        _lint_or_reject(out)                  # (before the stamp: a rejected library is neither stamped nor left under its name)
    with open(stamp, "w") as f:
        f.write(flags_digest(extra) + "\n")
    return out


def build_nan_clamp(force: bool = False, verbose: bool = False) -> str:
    """libmigan_hip_nanclamp.so: the same library without the NaN repair of lrelu_agc's clamp (Generator(nan_policy="clamp"): a NaN
    activation becomes -256 as in the reference's CUDA plugin; the default library propagates it as Tensor.clamp does)"""
    return build_variant("nanclamp", ["-DMIGAN_NAN_CLAMP"], force=force, verbose=verbose, lint=True)


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True,
                extra=[a for a in sys.argv[1:] if a.startswith("-D") or a.startswith("-R") or a.startswith("-save")]))
