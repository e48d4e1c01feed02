"""Dump per-kernel stats (count / avg / min / max / total) from a rocprofv3 rocpd sqlite DB as text."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = f"""select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0, max(d.end-d.start)/1000.0,
       sum(d.end-d.start)/1e6 from {disp} d join {sym} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc"""
print(f"{'kernel':72s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'total_ms':>9s}")
for r in cur.execute(q):
    print(f"{r[0][:72]:72s} {r[1]:6d} {r[2]:9.1f} {r[3]:9.1f} {r[4]:9.1f} {r[5]:9.2f}")
