"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) as a text table.
usage: prof_summary.py <dir> [rows] [--last-pass KERNEL | --timed-pass]
--last-pass KERNEL: the traced command ran the workload twice (one warm-up pass that also carries the GEMM autotuner's timing
launches, one timed pass); KERNEL is a kernel that closes a pass (stage 2's final k_gather_codebook): only kernels that start after
the first half of its occurrences are summarised, i.e. the timed pass alone.
--timed-pass: bench.py of round 2 runs a truncated warm-up pass and ONE timed pass; each pass initialises stage 2 exactly once
(k_scatter_final).  The timed pass starts after the last k_gather_codebook that precedes the second k_scatter_final."""
import glob
import re
import sqlite3
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
db = glob.glob(args[0] + "/*.db")[0]
nrows = int(args[1]) if len(args) > 1 else 45
c = sqlite3.connect(db)
t0 = 0
if "--last-pass" in sys.argv:
    mark = sys.argv[sys.argv.index("--last-pass") + 1]
    ends = [r[0] for r in c.execute("select end from kernels where name like ? order by start", (f"%{mark}%",))]
    t0 = ends[len(ends) // 2 - 1]
if "--timed-pass" in sys.argv:
    sf = [r[0] for r in c.execute("select start from kernels where name like '%k_scatter_final%' order by start")]
    if len(sf) >= 2:
        t0 = c.execute("select max(end) from kernels where name like '%k_gather_codebook%' and start < ?", (sf[-1],)).fetchone()[0] or 0
        w = c.execute("select min(start), max(end) from kernels where start > ?", (t0,)).fetchone()
        print(f"# timed pass only: {(w[1] - w[0]) / 1e9:.2f} s between its first and last kernel")
rows = c.execute("select name, count(*), sum(duration)/1e3, avg(duration)/1e3 from kernels where start > ? group by name order by 3 desc", (t0,)).fetchall()
tot = sum(r[2] for r in rows)


def short(n):
    n = re.sub(r"void rocprim::.*?(merge_sort_block_merge|radix_sort_block_sort|transform_impl|radix_sort_onesweep|histogram|scan).*", r"rocprim::\1<...>", n)
    return n[:100]


print(f"{'kernel':102s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
for r in rows[:nrows]:
    print(f"{short(r[0]):102s} {r[1]:7d} {r[2]:12.1f} {r[3]:10.2f} {100 * r[2] / tot:6.2f}")
print(f"TOTAL_us {tot:.1f}")
