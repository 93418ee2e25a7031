#!/usr/bin/env python3
"""Timeline of the LAST forward in a rocprofv3 kernel trace (rocpd SQLite): start offset, duration, queue, and the gaps in which
no kernel of ours was running.  usage: timeline.py trace.db [first_kernel_substring]"""
import re
import sqlite3
import sys


def main(path, first='stem_pool'):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in db.execute('pragma table_info(%s)' % kd)]
    qcol = 'queue_id' if 'queue_id' in cols else cols[0]
    rows = list(db.execute('select s.kernel_name, d.start, d.end, d.%s from %s d join %s s on d.kernel_id=s.id order by d.start' % (qcol, kd, ks)))
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    i0 = starts[-2] if len(starts) > 1 else starts[-1]
    i1 = starts[-1] if len(starts) > 1 else len(rows)
    seg = rows[i0:i1]
    t0 = seg[0][1]
    busy_end, idle, covered = t0, 0, 0
    for name, s, e, q in seg:
        short = re.sub(r'^_ZN\d*[a-z_]*\d*(_GLOBAL__N_1)?\d*', '', name)[:46]
        gap = max(0, s - busy_end)
        idle += gap
        print('%9.1f %8.1f q%-3s %s%s' % ((s - t0) / 1e3, (e - s) / 1e3, q, short, '   <gap %.1f>' % (gap / 1e3) if gap > 1500 else ''))
        busy_end = max(busy_end, e)
    print('forward span %.1f us, idle (no kernel running) %.1f us, sum of kernel durations %.1f us' %
          ((busy_end - t0) / 1e3, idle / 1e3, sum(e - s for _, s, e, _ in seg) / 1e3))


if __name__ == '__main__':
    main(*sys.argv[1:])
