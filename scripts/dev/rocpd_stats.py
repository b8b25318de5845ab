"""dev: per-kernel count / average duration from a rocprofv3 rocpd database (the default output format of rocprofv3 7.x)"""
import sqlite3, sys, glob
for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
    q = f"select s.kernel_name, count(*), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"
    for r in c.execute(q):
        print('%-90s %6d %9.1f us' % (r[0][:90], r[1], r[2] / 1e3))
