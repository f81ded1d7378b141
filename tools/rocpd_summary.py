#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite DB: per-kernel durations and PMC sums.
   python tools/rocpd_summary.py path/to/results.db [name-filter]"""
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
con = sqlite3.connect(db)
tabs = {r[0].rsplit('_', 5)[0]: r[0] for r in con.execute("select name from sqlite_master where type='table'")}


def T(n):
    return '"%s"' % tabs[n]


cols = [r[1] for r in con.execute(f'pragma table_info({T("rocpd_kernel_dispatch")})')]
kcols = [r[1] for r in con.execute(f'pragma table_info({T("rocpd_info_kernel_symbol")})')]
name_col = 'kernel_name' if 'kernel_name' in kcols else 'display_name'
rows = con.execute(f'select d.id, k.{name_col}, d.start, d.end, d.grid_size_x, d.workgroup_size_x, d.group_segment_size, '
                   f'd.private_segment_size from {T("rocpd_kernel_dispatch")} d join {T("rocpd_info_kernel_symbol")} k '
                   f'on d.kernel_id = k.id order by d.start').fetchall()
by = defaultdict(list)
info = {}
for did, name, s, e, gx, wx, lds, scr in rows:
    if flt and flt not in name:
        continue
    short = name.split('(')[0][:70]
    by[short].append((e - s) / 1e3)
    info[short] = (gx, wx, lds, scr)
print(f'{"kernel":72s} {"calls":>5s} {"avg_us":>10s} {"min_us":>10s} {"max_us":>10s}  grid wg lds scratch')
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f'{k:72s} {len(v):5d} {sum(v)/len(v):10.1f} {min(v):10.1f} {max(v):10.1f}  {info[k]}')
n = con.execute(f'select count(*) from {T("rocpd_pmc_event")}').fetchone()[0]
if n:
    pm = con.execute(f'select k.{name_col}, p.name, sum(e.value), count(*) from {T("rocpd_pmc_event")} e '
                     f'join {T("rocpd_info_pmc")} p on e.pmc_id = p.id '
                     f'join {T("rocpd_kernel_dispatch")} d on e.event_id = d.event_id '
                     f'join {T("rocpd_info_kernel_symbol")} k on d.kernel_id = k.id group by 1, 2').fetchall()
    nd = {k: len(v) for k, v in by.items()}
    print('\nPMC (sum over dispatches / number of dispatches):')
    for name, ctr, val, cnt in pm:
        if flt and flt not in name:
            continue
        short = name.split('(')[0][:70]
        print(f'  {short[:52]:52s} {ctr:28s} {val / nd[short]:16.0f}')
