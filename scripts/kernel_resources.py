#!/usr/bin/env python3
"""Register / LDS / occupancy table of one kernel-table slice, from hipcc's -Rpass-analysis=kernel-resource-usage
(no GPU needed).  usage: python scripts/kernel_resources.py G S [filter-substring]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mi-gan_amd", "csrc")


def demangle_args(sym: str) -> str:
    """_ZN5migan14sepconv_kernelILi0ELi128E...EEvNS_7SepArgsE -> template argument list"""
    m = re.search(r"I((?:L[ib][0-9n]+E)+)E", sym)
    if not m:
        return sym
    out = []
    for t, v in re.findall(r"L([ib])(n?[0-9]+)E", m.group(1)):
        out.append(("true" if v == "1" else "false") if t == "b" else v.replace("n", "-"))
    return ("wide" if "wide" in sym else "dwfir" if "dwfir" in sym else "sep") + "<" + ",".join(out) + ">"


def main():
    g, s = sys.argv[1], sys.argv[2]
    flt = sys.argv[3] if len(sys.argv) > 3 else ""
    src = "migan_hip.hip" if g == "main" else "migan_k_slice.hip"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
           "-c", src, "-o", "/tmp/_kr.o"] + os.environ.get("MIGAN_HIPCC_FLAGS", "").split()
    if g != "main":
        cmd += [f"-DMIGAN_SLICE_G={g}", f"-DMIGAN_SLICE_S={s}"]
    err = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    cur, rows = None, {}
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur:
                rows[cur][key] = int(m.group(1))
    print(f"{'kernel':70s} vgpr agpr scratch occ sgpr")
    for k, v in sorted(rows.items(), key=lambda kv: demangle_args(kv[0])):
        name = demangle_args(k)
        if flt in name:
            print(f"{name:70s} {v.get('vgpr', -1):4d} {v.get('agpr', -1):4d} {v.get('scratch', -1):7d} {v.get('occ', -1):3d} {v.get('sgpr', -1):4d}")


if __name__ == "__main__":
    main()
