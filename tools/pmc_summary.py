#!/usr/bin/env python
"""Per-kernel mean of one rocprofv3 PMC counter from a *_counter_collection.csv."""
import csv
import re
import sys
from collections import defaultdict

path, counter = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
with open(path) as f:
    for row in csv.DictReader(f):
        if row.get("Counter_Name") != counter:
            continue
        name = row.get("Kernel_Name", "")
        m = re.search(r"(chain_kernel|rtb_kernel|conv_kernel)<mmd::(?:Rtb|Chain)?Cfg<([^>]*)>", name)
        plain = name.replace("(anonymous namespace)::", "").replace("void ", "")
        key = f"{m.group(1)}<{m.group(2).replace(' ', '')}>" if m else re.sub(r"\(.*", "", plain)[:48]
        if "mconv" in key:                                   # the layered path's kernel: one line per launch shape
            key += f" grid {row.get('Grid_Size', '?')}"
        acc[key].append(float(row["Counter_Value"]))
print(f"# {counter}: mean per dispatch (raw counter units as reported by rocprofv3)")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:60s} n={len(v):5d} mean={sum(v) / len(v):14.1f}")
