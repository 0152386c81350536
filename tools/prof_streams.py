"""Per-queue occupancy of a rocprofv3 --kernel-trace run (rocpd sqlite): busy time (union of kernel intervals) of every queue / stream,
of the whole GPU, and the idle gaps -- tells whether a pass is bound by GPU work or by launch gaps / cross-stream waits.
usage: prof_streams.py <dir> [t0_fraction]   (t0_fraction: skip the first part of the trace, e.g. 0.3 to leave the warm-up pass out)"""
import glob
import sqlite3
import sys

db = glob.glob(sys.argv[1] + "/*.db")[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = next((k for k in ("stream_id", "queue_id", "stream", "queue") if k in cols), None)
lo, hi = c.execute("select min(start), max(end) from kernels").fetchone()
t0 = lo + (hi - lo) * (float(sys.argv[2]) if len(sys.argv) > 2 else 0.0)
rows = c.execute(f"select start, end, {qcol or '0'}, name from kernels where start >= ? order by start", (t0,)).fetchall()


def union(iv):
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    return tot + (cur_e - cur_s if cur_e is not None else 0)


wall = rows[-1][1] - rows[0][0] if rows else 0
wall = max(r[1] for r in rows) - rows[0][0]
print(f"columns: {cols}\nqueue column: {qcol}; {len(rows)} kernels, window {wall / 1e9:.3f} s")
print(f"GPU busy (union of all kernels) {union([(r[0], r[1]) for r in rows]) / 1e9:.3f} s; sum of kernel durations {sum(r[1] - r[0] for r in rows) / 1e9:.3f} s")
qs = sorted({r[2] for r in rows})
for q in qs:
    iv = [(r[0], r[1]) for r in rows if r[2] == q]
    print(f"queue {q}: {len(iv)} kernels, busy {union(iv) / 1e9:.3f} s, sum {sum(e - s for s, e in iv) / 1e9:.3f} s")
if len(qs) >= 2:
    main = max(qs, key=lambda q: sum(1 for r in rows if r[2] == q))
    a = [(r[0], r[1]) for r in rows if r[2] == main]
    b = [(r[0], r[1]) for r in rows if r[2] != main]
    both = union(a) + union(b) - union(sorted(a + b))
    print(f"main queue {main}: time with the other queue(s) also busy {both / 1e9:.3f} s")
