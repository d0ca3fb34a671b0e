"""Scan of the gfx950 code objects inside libmigan_hip.so for an instruction form that gives intermittently wrong results on
MI355X (profiles/r02_torgb_packed_f32_hazard.md): packed-fp32 FMA / add whose op_sel operand swizzle
makes the LOW result lane read the HIGH register of a source pair (`v_pk_fma_f32 ... op_sel:[0,1,0]`).  hipcc builds it when
it vectorises dot products with scalar operands; the CPU emulator executes the same source, not the ISA, and cannot see it.

`build.py` runs `check()` on every freshly linked library and fails the build on a hit; tests/test_isa_lint.py runs it on
whatever library the package would load.  Needs the ROCm LLVM tools (llvm-objcopy, clang-offload-bundler, llvm-objdump).
"""
from __future__ import annotations

import collections
import os
import re
import subprocess
import tempfile
from typing import Dict, List, Tuple

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
TOOLS = ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")


def available() -> bool:
    return all(os.path.exists(os.path.join(LLVM, t)) for t in TOOLS)


def unbundle(lib: str, tmp: str, disassemble: bool) -> List[Tuple[str, str]]:
    """[(code object path, disassembly or "")], one per translation unit (the .hip_fatbin section is a sequence of offload bundles)"""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    data = open(fat, "rb").read()
    offs = [m.start() for m in re.finditer(MAGIC, data)]
    if not offs:
        raise RuntimeError(f"no device code objects found in {lib}")
    out = []
    for k, o in enumerate(offs):
        end = offs[k + 1] if k + 1 < len(offs) else len(data)
        b, co = os.path.join(tmp, f"b{k}.bin"), os.path.join(tmp, f"b{k}.co")
        with open(b, "wb") as f:
            f.write(data[o:end])
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets={TARGET}", f"--input={b}", f"--output={co}"],
                       check=True)
        out.append((co, subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
                    if disassemble else ""))
    return out


def scan(lib: str, tmp: str) -> Dict[str, object]:
    """{"bad": Counter{(kernel, opcode): n}, "kernels": set of symbols, "mfma": count of fp16 matrix instructions seen}"""
    bad: collections.Counter = collections.Counter()
    kernels = set()
    mfma = 0
    for _, text in unbundle(lib, tmp, disassemble=True):
        cur = None
        for line in text.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1)
                kernels.add(cur)
                continue
            mfma += "v_mfma_f32_32x32x16_f16" in line
            m = re.search(r"\b(v_pk_(?:fma|add)_f32)\b.*\bop_sel:\[([01,]+)\]", line)
            if m and "1" in m.group(2):
                bad[(cur, m.group(1))] += 1
    return {"bad": bad, "kernels": kernels, "mfma": mfma}


def check(lib: str) -> Dict[str, object]:
    """raise RuntimeError if any kernel of `lib` contains the hazardous form"""
    with tempfile.TemporaryDirectory() as tmp:
        r = scan(lib, tmp)
    bad = r["bad"]
    if bad:
        raise RuntimeError("hazardous packed-fp32 instruction form (low-lane op_sel on v_pk_fma_f32 / v_pk_add_f32) in: "
                           + ", ".join(f"{k[:80]} ({op} x{n})" for (k, op), n in bad.most_common(8)))
    return r
