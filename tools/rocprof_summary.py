#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) result into the per-kernel summary committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_r1a/bench_results.db > profiles/r01_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name)            # drop the argument list
    name = name.replace("void ", "")
    if len(name) > 90:
        name = name[:87] + "..."
    return name


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print(f"# source: {path}")
    print(f"# rocprofv3 --kernel-trace --stats ; durations in microseconds")
    print(f"{'kernel':<92} {'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>7}")
    for name, calls, total, avg, pct in rows:
        print(f"{short(name):<92} {calls:>6} {total:>12.1f} {avg:>10.2f} {pct:>7.2f}")
    # per-kernel register / LDS footprint of the library's kernels
    print("\n# dispatch footprint (one row per distinct kernel of libsinddm_hip.so)")
    q = ("select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), "
         "max(workgroup_x), min(duration), max(duration) from kernels where name like '%sinddm%' group by name")
    print(f"{'kernel':<60} {'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds_B':>7} {'grid_x':>9} {'wg':>5} {'min_ns':>10} {'max_ns':>10}")
    for r in cur.execute(q):
        print(f"{short(r[0]):<60} {r[1]:>5} {r[2]:>5} {r[3]:>5} {r[4]:>7} {r[5]:>9} {r[6]:>5} {r[7]:>10} {r[8]:>10}")


if __name__ == "__main__":
    main(sys.argv[1])
