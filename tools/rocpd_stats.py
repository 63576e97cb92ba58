"""Summarise a rocprofv3 rocpd database (kernel-trace) as a per-kernel stats table (CSV on stdout).

    python tools/rocpd_stats.py gpurun_out/prof_r1/bench_results.db > profiles/r1_kernel_stats.csv
Same columns as `rocprofv3 --stats` kernel_stats.csv: Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, (end - start) from kernels").fetchall()
    agg = {}
    for n, d in rows:
        a = agg.setdefault(short(n), [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('"%s",%d,%d,%.1f,%.3f,%d,%d' % (n.replace('"', "'"), a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot, a[2], a[3]))


if __name__ == "__main__":
    main(sys.argv[1])
