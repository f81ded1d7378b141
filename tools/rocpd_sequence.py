#!/usr/bin/env python3
"""Print the kernel dispatch sequence of ONE steady-state step from a rocprofv3 rocpd DB: everything between the
last two launches of the kernel whose name contains `anchor`.   python tools/rocpd_sequence.py db anchor"""
import sqlite3
import sys
con = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2]
tabs = {r[0].rsplit('_', 5)[0]: r[0] for r in con.execute("select name from sqlite_master where type='table'")}
rows = con.execute(f'select k.kernel_name, d.start, d.end from "{tabs["rocpd_kernel_dispatch"]}" d join '
                   f'"{tabs["rocpd_info_kernel_symbol"]}" k on d.kernel_id = k.id order by d.start').fetchall()
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
# the replayed train step, not the back-to-back eager calls around it: among consecutive anchor launches take the last pair
# whose dispatch count is the most common one above 1 (falls back to the last pair)
from collections import Counter
pairs = [(idx[k], idx[k + 1]) for k in range(len(idx) - 1)]
cnt = Counter(b - a for a, b in pairs if b - a > 1)      # (the folded train step is two dispatches; bare loops are one)
if cnt:
    mode = cnt.most_common(1)[0][0]
    a, b = [pr for pr in pairs if pr[1] - pr[0] == mode][-1]
else:
    a, b = idx[-2], idx[-1]
t_prev_end = rows[a][2]
busy = 0.0
print(f'{b - a} dispatches per step; step period {(rows[b][1] - rows[a][1]) / 1e3:.1f} us')
for name, s, e in rows[a:b]:
    short = name.split('(')[0]
    short = short[-60:]
    print(f'  gap {max(0, s - t_prev_end) / 1e3:7.1f} us   run {(e - s) / 1e3:8.1f} us   {short}')
    busy += (e - s) / 1e3
    t_prev_end = e
print(f'sum of kernel run times {busy:.1f} us')
