import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start) from kernels group by name order by 3 desc").fetchall()
for n, c, ns in rows[:40]:
    print(f"{c:5d} {ns/c/1e3:8.1f} us  {n[:400]}")
