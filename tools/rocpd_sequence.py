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
