#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table (markdown or csv).
    python tools/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/rNN_kernel_stats.md"""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"(chain_kernel|conv_kernel)<mmd::(Chain)?Cfg<([^>]*)>", name)
    if m:
        return f"{m.group(1)}<{m.group(2) or ''}Cfg<" + m.group(3).replace(" ", "") + ">>"
    return re.sub(r"\(.*", "", name)[:90]


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"# rocprofv3 --kernel-trace --stats summary ({path})\n")
    print(f"total kernel time {tot / 1e6:.3f} ms over {sum(a[0] for a in agg.values())} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{k}` | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.2f} | {a[2] / 1e3:.2f} | {a[3] / 1e3:.2f} | "
              f"{100.0 * a[1] / tot:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1])
