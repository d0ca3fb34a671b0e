"""ISA lint of the built product library (no GPU needed: the device code objects inside libmigan_hip.so are disassembled).

Round 2 found an instruction form that gives intermittently wrong results on MI355X (profiles/r02_torgb_packed_f32_hazard.md and
DESIGN.md section 5.7): packed-fp32 FMA / add whose op_sel operand swizzle makes the LOW result lane read the HIGH register of a
source pair (`v_pk_fma_f32 ... op_sel:[0,1,0]`).  hipcc produces it when it SLP-vectorises dot products with scalar operands (the
fused / un-fused ToRGB tails; a hoisted FromRGB loop did it again and was wrong on 1-2 % of the first layer's outputs, different
elements every launch, while the CPU emulator -- which executes the same source -- was clean).  The shipped kernels avoid the form
(scalar FMA chains with optimisation barriers where the vectoriser would build it); this test keeps it that way by scanning every
kernel of every code object.  `v_pk_mul_f32` with op_sel (used by the activation code of every kernel, run billions of times in the
GPU parity and determinism tests) is not part of the pattern."""
import importlib
import os
import re
import subprocess

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"


def _lint():
    return importlib.import_module("mi-gan_amd.isa_lint")


def code_objects(lib, tmp):
    """paths of the gfx950 code objects inside the library, one per translation unit"""
    return [co for co, _ in _lint().unbundle(lib, tmp, disassemble=False)]


@pytest.mark.skipif(not os.path.exists(f"{LLVM}/llvm-objdump"), reason="needs the ROCm LLVM tools")
def test_no_packed_fp32_fma_or_add_with_low_lane_operand_swizzle(tmp_path):
    """the same scan mi-gan_amd/build.py runs after every link (a hit fails the build), here on whatever the package would load"""
    pkg = importlib.import_module("mi-gan_amd")
    lib = pkg.library_path()
    if not os.path.exists(lib):
        importlib.import_module("mi-gan_amd.build").build()
    r = _lint().scan(lib, str(tmp_path))
    bad, kernels, mfma = r["bad"], r["kernels"], r["mfma"]
    # the scan really saw the product kernels
    assert any("sepconv_kernel" in k for k in kernels) and any("cm_conv_kernel" in k for k in kernels) and mfma > 1000
    assert not bad, "hazardous packed-fp32 instruction form in: " + ", ".join(f"{k[:80]} ({op} x{n})" for (k, op), n in bad.most_common(8))


def test_build_refuses_measurement_builds_as_fresh(tmp_path, monkeypatch):
    """a library compiled with extra flags (-DMIGAN_PHASE_PROF) must not pass for the product on the next plain build()"""
    b = importlib.import_module("mi-gan_amd.build")
    assert b.flags_digest(()) != b.flags_digest(("-DMIGAN_PHASE_PROF",))
    if os.path.exists(b.OUT) and os.path.exists(b.STAMP):
        assert b.is_fresh(()) in (True, False)
        assert not b.is_fresh(("-DMIGAN_PHASE_PROF",))


def kernel_resources(lib, tmp):
    """{demangled-ish kernel name: dict(vgpr, agpr, scratch, sgpr_spill, vgpr_spill)} from the code objects' metadata notes"""
    res = {}
    for co in code_objects(lib, tmp):
        txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
        cur = {}
        for line in txt.split("\n"):
            m = re.match(r"\s*-?\s*\.(agpr_count|name|private_segment_fixed_size|sgpr_spill_count|vgpr_count|vgpr_spill_count):\s*(\S+)", line)
            if not m:
                continue
            if m.group(1) == "agpr_count" and cur.get("name"):
                res[cur["name"]] = cur
                cur = {}
            cur[m.group(1)] = m.group(2) if m.group(1) == "name" else int(m.group(2))
        if cur.get("name"):
            res[cur["name"]] = cur
    return res


def _targs(sym):
    """template arguments of a mangled migan kernel symbol as a tuple of ints (bools as 0/1)"""
    m = re.search(r"I((?:L[ib]n?[0-9]+E)+)E", sym)
    return tuple(int(v.replace("n", "-")) for _, v in re.findall(r"L([ib])(n?[0-9]+)E", m.group(1))) if m else ()


@pytest.mark.skipif(not os.path.exists(f"{LLVM}/llvm-readelf"), reason="needs the ROCm LLVM tools")
def test_occupancy_budgets_of_the_default_plan_kernels(tmp_path):
    """Round 2's measured rule: a tile gains 18-21 % from a third workgroup per CU only where it fits the budget
    WITHOUT spilling.  The kernels the default plans rely on for that must stay inside 168 VGPRs (three waves per SIMD; 128 for
    four) with no scratch, whatever a later edit or compiler does to them; the Co-Mod-GAN four-phase tile must keep two waves."""
    pkg = importlib.import_module("mi-gan_amd")
    res = kernel_resources(pkg.library_path(), str(tmp_path))
    sep = {_targs(k): v for k, v in res.items() if "sepconv_kernel" in k and "narrow_" not in k}
    dwf = {_targs(k): v for k, v in res.items() if "dwfir_kernel" in k}
    cmc = {_targs(k): v for k, v in res.items() if "cm_conv_kernel" in k}
    assert len(sep) > 100 and len(dwf) == 10 and len(cmc) >= 8

    def within(v, vgpr):
        return v["vgpr_count"] + v["agpr_count"] <= vgpr and v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] == 0

    # (MODE, MT, NT, KC, FROMRGB, NI, MINW, MAING, PERSIST, GEMMV, TORGB, STV)
    for gemmv, stv in ((2, 0), (3, 1), (3, 2)):
        for torgb in (0, 1):
            plain = sep[(0, 128, 64, 32, 0, 6, 3, 1, 0, gemmv, torgb, stv)]          # Cin = 64 plain / ToRGB tile, two unrolled chunks
            assert within(plain, 168 if stv == 0 else 128), plain
        assert within(sep[(0, 128, 64, 32, 1, 6, 3, 1, 0, gemmv, 0, stv)], 168)       # fused-FromRGB tile
        if stv:
            assert within(sep[(2, 128, 64, 32, 0, 6, 2, 1, 0, gemmv, 0, stv)], 168)   # 16-bit FIR-up tile, one tile per workgroup
    for (ni, maing, stv), v in dwf.items():
        if maing:
            assert within(v, 168), (ni, maing, stv, v)                               # depthwise + FIR-down: three workgroups per CU
    assert within(cmc[(64, 32, 6, 1, 2, 1)], 256)                                     # four-phase transposed conv: two waves per SIMD
    # no main-geometry kernel of the f16x2 / f16 GEMM variants (what the 256 / 512 plans launch) runs with scratch; known to
    # spill and not in any default plan: the 3-workgroup FIR-up tile and the 16-channel-chunk tiles at 3-4 workgroups per CU
    spilled = sorted(k for k, v in sep.items() if v["private_segment_fixed_size"] and k[7] == 1 and k[9] in (2, 3) and k[3] == 32
                     and not (k[0] == 2 and k[6] == 3))
    assert spilled == [], spilled


@pytest.mark.skipif(not os.path.exists(f"{LLVM}/llvm-readelf"), reason="needs the ROCm LLVM tools")
def test_pipelined_kernels_fit_their_workgroup(tmp_path):
    """sepconv_pipe_kernel / sepconv_pipedown_kernel run ONE workgroup of 12 or 16 waves per CU: 3 or 4 waves per SIMD, i.e. at most 168
    or 128 registers per lane.  No instantiation may spill (a spilling form was measured 1.7x slower: profiles/r04_pipe_layers.txt)."""
    pkg = importlib.import_module("mi-gan_amd")
    res = kernel_resources(pkg.library_path(), str(tmp_path))
    pipe = {_targs(k): v for k, v in res.items() if "sepconv_pipe_kernel" in k}          # (MODE, NT, CIN, FROMRGB, TORGB, R, NA)
    down = {_targs(k): v for k, v in res.items() if "sepconv_pipedown_kernel" in k}      # (NT, CIN, R, NA, NB)
    assert len(pipe) >= 9 and len(down) >= 5
    for table in (pipe, down):
        for targs, v in table.items():
            waves = targs[-1] + 8 if table is pipe else targs[-2] + targs[-1]
            cap = 168 if waves == 12 else 128
            # round 6: the FIR-up form (MODE 2) sits exactly at its 168-register cap, and the NaN test of the clamp (two SGPR pairs per four
            # values) costs it ONE lane constant kept in scratch: written once before the tile loop, reloaded once per TILE (checked in the ISA:
            # no scratch access inside the K loop).  Anything more is accumulator / window traffic again and fails here.
            scratch_cap = 8 if (table is pipe and targs[0] == 2) else 0
            assert v["vgpr_count"] + v["agpr_count"] <= cap and v["private_segment_fixed_size"] <= scratch_cap and v["vgpr_spill_count"] <= scratch_cap // 8, (targs, v)


@pytest.mark.skipif(not os.path.exists(f"{LLVM}/llvm-readelf"), reason="needs the ROCm LLVM tools")
def test_wide2_kernel_fits_its_workgroup(tmp_path):
    """sepconv_wide2_kernel: one 12-wave workgroup per CU, 128 accumulator registers per MFMA wave -- it must stay inside the 168
    registers three waves per SIMD leave, without scratch (the first form spilled 200 registers around the accumulator blocks)."""
    pkg = importlib.import_module("mi-gan_amd")
    res = kernel_resources(pkg.library_path(), str(tmp_path))
    w2 = {k: v for k, v in res.items() if "sepconv_wide2_kernel" in k}
    assert len(w2) >= 1
    for k, v in w2.items():
        # (the 32-channel-chunk form keeps three lane constants -- the lane number and a store offset -- in scratch and reloads them once per
        # tile, outside the MFMA steps: 12 bytes; anything more would be accumulator traffic again)
        assert v["vgpr_count"] + v["agpr_count"] <= 168 and v["private_segment_fixed_size"] <= 16, (k, v)
