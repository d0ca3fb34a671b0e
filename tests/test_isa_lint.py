"""ISA lint of the built product library (no GPU needed: the device code objects inside libmigan_hip.so are disassembled).

Round 2 found an instruction form that gives intermittently wrong results on MI355X (profiles/r02_torgb_packed_f32_hazard.md and
DESIGN.md section 5.7): packed-fp32 FMA / add whose op_sel operand swizzle makes the LOW result lane read the HIGH register of a
source pair (`v_pk_fma_f32 ... op_sel:[0,1,0]`).  hipcc produces it when it SLP-vectorises dot products with scalar operands (the
fused / un-fused ToRGB tails; a hoisted FromRGB loop did it again and was wrong on 1-2 % of the first layer's outputs, different
elements every launch, while the CPU emulator -- which executes the same source -- was clean).  The shipped kernels avoid the form
(scalar FMA chains with optimisation barriers where the vectoriser would build it); this test keeps it that way by scanning every
kernel of every code object.  `v_pk_mul_f32` with op_sel (used by the activation code of every kernel, run billions of times in the
GPU parity and determinism tests) is not part of the pattern."""
import collections
import importlib
import os
import re
import subprocess

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def device_disassembly(lib, tmp):
    """one disassembly per translation unit of the library (its .hip_fatbin section is a sequence of offload bundles)"""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    data = open(fat, "rb").read()
    offs = [m.start() for m in re.finditer(MAGIC, data)]
    assert offs, "no device code objects found in the library"
    out = []
    for k, o in enumerate(offs):
        end = offs[k + 1] if k + 1 < len(offs) else len(data)
        b, co = os.path.join(tmp, f"b{k}.bin"), os.path.join(tmp, f"b{k}.co")
        with open(b, "wb") as f:
            f.write(data[o:end])
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets={TARGET}", f"--input={b}", f"--output={co}"],
                       check=True)
        out.append(subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout)
    return out


@pytest.mark.skipif(not os.path.exists(f"{LLVM}/llvm-objdump"), reason="needs the ROCm LLVM tools")
def test_no_packed_fp32_fma_or_add_with_low_lane_operand_swizzle(tmp_path):
    pkg = importlib.import_module("mi-gan_amd")
    lib = pkg.library_path()
    if not os.path.exists(lib):
        importlib.import_module("mi-gan_amd.build").build()
    bad = collections.Counter()
    kernels = set()
    mfma = 0
    for text in device_disassembly(lib, str(tmp_path)):
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
    # the scan really saw the product kernels
    assert any("sepconv_kernel" in k for k in kernels) and any("cm_conv_kernel" in k for k in kernels) and mfma > 1000
    assert not bad, "hazardous packed-fp32 instruction form in: " + ", ".join(f"{k[:80]} ({op} x{n})" for (k, op), n in bad.most_common(8))
