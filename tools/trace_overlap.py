import sqlite3, sys
c = sqlite3.connect(sys.argv[1]).cursor()
rows = list(c.execute("select name,start,end from kernels where name like '%render_kernel%' order by start"))
t0 = rows[0][1]
prev_end = None
for name, s, e in rows[int(sys.argv[2]) if len(sys.argv) > 2 else -14:]:
    gap = "" if prev_end is None else f" start-prev_end {(s - prev_end)/1e3:8.1f} us"
    print(f"{(s-t0)/1e6:9.3f} ms .. {(e-t0)/1e6:9.3f} ms  dur {(e-s)/1e6:7.3f} ms{gap}")
    prev_end = e
