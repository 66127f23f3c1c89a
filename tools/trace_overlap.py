"""Read a rocprofv3 rocpd db; report per-stream kernel overlap for the conv kernels."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
rows = c.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
rows = [r for r in rows if 'conv_kernel' in r[0] or 'ddpm' in r[0]]
t0, t1 = rows[0][1], max(r[2] for r in rows)
busy = 0; cur_s, cur_e = None, None
for _, s, e, *_ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(r[2] - r[1] for r in rows)
print(f"span {(t1 - t0) / 1e6:.2f} ms, union busy {busy / 1e6:.2f} ms, sum of durations {tot / 1e6:.2f} ms, overlap factor {tot / busy:.2f}")
print("queues", sorted(set(r[3] for r in rows)), "streams", sorted(set(r[4] for r in rows)))
for r in rows[2000:2012]:
    print(r[0][:60], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[4])
