#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 results .db (kernel-trace): calls, avg us, total us."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, sum(d.end-d.start)/1000.0 "
     f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc limit {int(sys.argv[2]) if len(sys.argv) > 2 else 25}")
tot = list(cur.execute(f"select sum(end-start)/1000.0 from {kd}"))[0][0]
for name, n, avg, total in cur.execute(q):
    print(f"{name[:100]:100s} calls={n:5d} avg_us={avg:10.1f} pct={100*total/tot:6.2f}")
