"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) as a text table.  usage: prof_summary.py <dir> [rows]"""
import glob
import re
import sqlite3
import sys

db = glob.glob(sys.argv[1] + "/*.db")[0]
c = sqlite3.connect(db)
rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()


def short(n):
    n = re.sub(r"void rocprim::.*?(merge_sort_block_merge|radix_sort_block_sort|transform_impl|radix_sort_onesweep|histogram|scan).*", r"rocprim::\1<...>", n)
    return n[:100]


print(f"{'kernel':102s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    print(f"{short(r[0]):102s} {r[1]:7d} {r[2]:12.1f} {r[3]:10.2f} {r[4]:6.2f}")
print(f"TOTAL_us {sum(r[2] for r in rows):.1f}")
