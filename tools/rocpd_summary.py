#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the text table we keep under profiles/.

usage: tools/rocpd_summary.py gpurun_out/prof/r01_results.db > profiles/r01_kernel_stats.txt
Equivalent of `rocprofv3 --kernel-trace --stats` kernel_stats.csv plus the per-dispatch resource
columns (VGPRs, SGPRs, scratch, LDS) and PMC counter averages when the db holds any.
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start),"
        " max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(scratch_size), max(lds_size),"
        " max(grid_x), max(workgroup_x) from kernels group by name order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# source: {path}")
    print("# kernel | calls | total_ms | avg_us | min_us | max_us | pct | vgpr | agpr | sgpr | scratch_B | lds_B | grid_x | wg_x")
    for r in rows:
        name = r[0].split("(")[0]
        if len(name) > 70:
            name = name[:67] + "..."
        print(f"{name} | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e3:.1f} | {r[4]/1e3:.1f} | {r[5]/1e3:.1f} | "
              f"{100*r[2]/total:.2f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} | {r[12]}")
    try:
        pmc = c.execute(
            "select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection"
            " group by kernel_name, counter_name").fetchall()
    except sqlite3.Error:
        pmc = []
    if pmc:
        print("\n# PMC counters: kernel | counter | dispatches | avg_per_dispatch | sum")
        for r in pmc:
            print(f"{r[0].split('(')[0]} | {r[1]} | {r[2]} | {r[3]:.6g} | {r[4]:.6g}")


if __name__ == "__main__":
    main(sys.argv[1])
