#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (ROCm 7.2 default output of
`rocprofv3 --kernel-trace --stats`) into the per-kernel table `--stats` would print, plus a
per-dispatch-shape table (kernel x grid size = one network layer).  Usage:
    python scripts/rocpd_summary.py gpurun_out/prof/r01_results.db > profiles/<name>.md
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print(f"# rocprofv3 kernel-trace summary of `{path.split('/')[-1]}`\n")
    print("## per kernel (KERNEL_DISPATCH stats)\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    tot = cur.execute("select sum(duration) from kernels").fetchone()[0]
    for name, n, s, a, mn, mx in cur.execute(
            "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc"):
        print(f"| `{name}` | {n} | {s/1e6:.3f} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*s/tot:.2f} |")
    print("\n## per dispatch shape (one row = one layer of the forward, in launch order)\n")
    print("| kernel | workgroups | LDS B | arch VGPR | accum VGPR | SGPR | calls | avg us | min us |")
    print("|---|---|---|---|---|---|---|---|---|")
    for r in cur.execute("select name, grid_x/workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, count(*), avg(duration), min(duration) "
                         "from kernels where name like '%migan%' group by name, grid_x order by min(start)"):
        print(f"| `{r[0]}` | {r[1]} | {r[2]} | {r[3]} | {r[4]} | {r[5]} | {r[6]} | {r[7]/1e3:.1f} | {r[8]/1e3:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
