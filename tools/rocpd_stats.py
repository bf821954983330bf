"""Per-kernel totals (or, with --seq PATTERN, the matching dispatches in launch order) from a rocprofv3 rocpd database
(the default output of `rocprofv3 --kernel-trace` on this image): python3 tools/rocpd_stats.py results.db [--seq radius]"""
import sqlite3, sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, grid_x, workgroup_x, lds_size, vgpr_count, scratch_size from kernels order by start"))
if len(sys.argv) > 3 and sys.argv[2] == "--seq":
    t0 = rows[0][1]
    for r in rows:
        if sys.argv[3] in r[0]:
            print("%10.2f ms %10.1f us grid %9d wg %4d lds %6d vgpr %3d scr %4d %s" % ((r[1] - t0) / 1e6, (r[2] - r[1]) / 1e3, r[3], r[4], r[5], r[6], r[7], r[0][:80]))
else:
    tot = {}
    for r in rows:
        t = tot.setdefault(r[0], [0, 0.0])
        t[0] += 1
        t[1] += (r[2] - r[1]) / 1e3
    allt = sum(v[1] for v in tot.values())
    print("kernel,calls,total_us,avg_us,percent")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print('"%s",%d,%.3f,%.3f,%.3f' % (k, v[0], v[1], v[1] / v[0], 100 * v[1] / allt))
