"""dev: the last N records (kernels and memory copies) of a rocprofv3 rocpd database in time order, with the gaps between them"""
import sqlite3, sys
db, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 24
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = [(r[0], r[1], r[2][:60]) for r in c.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id")]
mc = [t for t in tabs if 'memory_copy' in t]
if mc:
    cols = [r[1] for r in c.execute(f"pragma table_info({mc[0]})")]
    rows += [(r[0], r[1], 'COPY %d B' % r[2]) for r in c.execute(f"select start, end, size from {mc[0]}")]
rows.sort()
prev = None
for st, en, name in rows[-n:]:
    print('%9.1f us  +gap %6.1f  dur %6.1f  %s' % ((st - rows[-n][0]) / 1e3, (st - prev) / 1e3 if prev else 0.0, (en - st) / 1e3, name))
    prev = en
