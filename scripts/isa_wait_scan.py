"""Where the `s_waitcnt vmcnt` of a kernel sit relative to its global loads and stores (no GPU needed: the code objects inside libmigan_hip.so are
disassembled).  `vmcnt` retires in issue order, so a load that is consumed after a store of the same wave waits for that store: the pattern
`S4 L8 w0` (four stores, eight loads, vmcnt(0)) in a loop is a store drain.  DESIGN.md 11.2 and profiles/LOG.md (round 6) list what this scan found.

usage: python scripts/isa_wait_scan.py [substring of the mangled kernel symbol ...]      (default: the kernels of the migan-512 forward)
legend: Ln = n global / buffer loads in a row, Sn = n stores, Dn = n LDS-DMAs (buffer_load ... lds), wN = s_waitcnt vmcnt(N), | = s_barrier.
The order is the order of the code in memory, not of execution: loops and the two wave groups of the pipelined kernels appear one after the other."""
import collections
import importlib
import os
import re
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEFAULT = ["sepconv_pipe_kernel", "sepconv_pipedown_kernel", "sepconv_wide2_kernel", "sepconv_wide_kernelILb0ELi0ELb0ELb1ELb1", "sepconv_wide_kernelILb1ELi0ELb0ELb1ELb1",
           "sepconv_kernelILi2ELi128ELi128ELi32ELb0ELi6ELi2ELb1ELb0ELi2ELb0ELi0E", "dwfir_kernelILi7ELb1ELi0E"]


def tokens(body):
    seq = []
    for l in body:
        if l.startswith(("global_store", "buffer_store")):
            seq.append("S")
        elif l.startswith("buffer_load") and " lds" in l:
            seq.append("D")
        elif l.startswith(("global_load", "buffer_load")):
            seq.append("L")
        elif l.startswith("s_waitcnt") and "vmcnt" in l:
            seq.append("w" + re.search(r"vmcnt\((\d+)\)", l).group(1))
        elif l.startswith("s_barrier"):
            seq.append("|")
    out, i = [], 0
    while i < len(seq):
        j = i
        while j < len(seq) and seq[j] == seq[i] and seq[i] in "SLD":
            j += 1
        out.append(f"{seq[i]}{j - i}" if j > i else seq[i])
        i = max(j, i + 1)
    return " ".join(out)


def main():
    want = sys.argv[1:] or DEFAULT
    lint = importlib.import_module("mi-gan_amd.isa_lint")
    pkg = importlib.import_module("mi-gan_amd")
    with tempfile.TemporaryDirectory() as tmp:
        for _, text in lint.unbundle(pkg.library_path(), tmp, disassemble=True):
            cur, bodies = None, collections.defaultdict(list)
            for line in text.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1)
                elif cur and "\t" in line:
                    bodies[cur].append(line.split("\t")[1].strip())
            for k, b in bodies.items():
                if any(w in k for w in want):
                    print(k)
                    print("   ", tokens(b))


if __name__ == "__main__":
    main()
