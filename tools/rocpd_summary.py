#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (--kernel-trace --stats) as text: per-kernel totals
(the `top_kernels` view) and every dispatch of the render kernels.  Usage:
    python tools/rocpd_summary.py gpurun_out/prof_x/x_results.db > profiles/rNN_x_kernel_stats.txt"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path).cursor()
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print("# name | calls | total_ns | average_ns | percent")
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(" | ".join(str(x) for x in r))
    print("\n# render-kernel dispatches: name | duration_ns | grid_x | workgroup_x | lds_bytes | vgpr(alloc granules) | sgpr | scratch")
    q = ("select name,duration,grid_x,workgroup_x,lds_size,vgpr_count,sgpr_count,scratch_size from kernels "
         "where name like '%render_kernel%' or name like '%trace_samples%' order by start")
    for r in c.execute(q):
        print(" | ".join(str(x) for x in r))


if __name__ == "__main__":
    main(sys.argv[1])
