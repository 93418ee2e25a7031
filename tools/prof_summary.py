#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) of a rocprofv3 rocpd SQLite result (kernel trace)."""
import re
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in db.execute('pragma table_info(%s)' % kd)]
    scol = [r[1] for r in db.execute('pragma table_info(%s)' % ks)]
    name_col = 'kernel_name' if 'kernel_name' in scol else 'display_name'
    q = 'select s.%s, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d join %s s on d.kernel_id=s.id group by s.%s order by 3 desc' % (name_col, kd, ks, name_col)
    rows = list(db.execute(q))
    tot = sum(r[2] for r in rows)
    print('%-78s %8s %12s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%'))
    for name, n, t, mn, mx in rows[:top]:
        short = re.sub(r'\(anonymous namespace\)::', '', name)
        short = re.sub(r'\(.*\)$', '', short)[:78]
        print('%-78s %8d %12.1f %10.2f %10.2f %10.2f %6.2f' % (short, n, t / 1e3, t / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
    print('%-78s %8d %12.1f' % ('TOTAL', sum(r[1] for r in rows), tot / 1e3))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
