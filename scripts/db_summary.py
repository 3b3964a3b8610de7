#!/usr/bin/env python3
"""Per-kernel time summary of a rocprofv3 (rocpd sqlite) kernel trace:  python scripts/db_summary.py <dir-or-db> [steps] [top]"""
import collections, glob, re, sqlite3, sys
path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
f = path if path.endswith(".db") else glob.glob(path + "/*/*_results.db")[0]
c = sqlite3.connect(f)
rows = c.execute("select name, start, end from kernels").fetchall()
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in rows:
    agg[n][0] += 1; agg[n][1] += (e - s)
tot = sum(v[1] for v in agg.values())
print("total ms/step %.3f   kernels/step %.1f" % (tot / 1e6 / steps, len(rows) / steps))
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%8.3f ms %6.1f  %7.1f us  %s" % (v[1] / 1e6 / steps, v[0] / steps, v[1] / v[0] / 1e3, re.sub(r"\s+", " ", n)[:118]))
