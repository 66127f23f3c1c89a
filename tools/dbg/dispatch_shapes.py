"""Kernel time of a rocprofv3 --kernel-trace run by (kernel, grid, workgroup size): dispatch_shapes.py <results.db>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = (f"select s.kernel_name, d.grid_size_x, d.grid_size_y, d.workgroup_size_x, count(*), avg(d.end-d.start) from {kd} d join {ks} s "
     f"on d.kernel_id=s.id group by 1,2,3,4 order by 1, 2")
tot = 0.0
for name, gx, gy, wg, cnt, avg in c.execute(q):
    short = name.replace("_ZN3mmd12_GLOBAL__N_1", "").replace(".kd", "")[:48]
    print(f"{short:48s} wgs {gx // wg:>6} x {gy:>2}  calls {cnt:>5}  avg {avg / 1e3:8.1f} us  total {cnt * avg / 1e6:8.2f} ms")
    tot += cnt * avg / 1e6
print(f"total {tot:.2f} ms")
