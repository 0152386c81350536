import sqlite3, sys, glob, collections
db = glob.glob(sys.argv[1] + '/*.db')[0]
c = sqlite3.connect(db)
tabs=[r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
v = [t for t in tabs if 'counter' in t.lower() or 'pmc' in t.lower()]
print(v[:12])
for t in v[:12]:
    try:
        cur=c.execute(f"select * from {t} limit 2"); print(t,[d[0] for d in cur.description]); print(cur.fetchall()[:2])
    except Exception as e: print(t, e)
