#!/usr/bin/env python3
"""Per-kernel PMC counter averages from a rocprofv3 rocpd SQLite result."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2] if len(sys.argv) > 2 else 'conv_igemm'
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
pe, pi, kd, ks = T('rocpd_pmc_event'), T('rocpd_info_pmc'), T('rocpd_kernel_dispatch'), T('rocpd_info_kernel_symbol')
pec = [r[1] for r in db.execute('pragma table_info(%s)' % pe)]
kdc = [r[1] for r in db.execute('pragma table_info(%s)' % kd)]
evcol = 'event_id' if 'event_id' in pec else pec[1]
q = ("select s.kernel_name, i.name, avg(e.value), count(*) from %s e join %s i on e.pmc_id=i.id join %s d on e.%s=d.event_id "
     "join %s s on d.kernel_id=s.id where s.kernel_name like ? group by s.kernel_name, i.name") % (pe, pi, kd, evcol, ks)
rows = list(db.execute(q, ('%' + pat + '%',)))
by = {}
for kn, cn, v, n in rows:
    by.setdefault(re.sub(r'\(.*', '', kn)[:60], {})[cn] = (v, n)
for kn, d in by.items():
    print(kn)
    for cn, (v, n) in sorted(d.items()):
        print('   %-32s %16.1f  (n=%d)' % (cn, v, n))
