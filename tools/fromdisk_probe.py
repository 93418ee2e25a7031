"""Where a decode worker's time goes (per image, one process) and what the box grants: python tools/fromdisk_probe.py"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'helpers'))
from fake_split import write_split  # noqa: E402
from dir_amd.apps import dataset as DS  # noqa: E402
from dir_amd.apps import jpeg as AJ  # noqa: E402

for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    if os.path.exists(f):
        print(f, open(f).read().strip())
print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())
with tempfile.TemporaryDirectory() as d:
    write_split(d, 256, seed=1)
    ds = DS.InterHandSplit(d)
    row = np.zeros(AJ.record_bytes(256), np.uint8)

    def t(f, n=256, reps=3):
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            for i in range(n):
                f(i)
            best = min(best, (time.perf_counter() - t0) / n)
        return best * 1e6
    print('anno (open + unpickle + pack)      %.0f us' % t(ds.anno))
    print('read file only                     %.0f us' % t(lambda i: open(ds.img_path(i), 'rb').read()))
    print('file_to_record (read + Huffman)    %.0f us' % t(lambda i: AJ.file_to_record(ds.img_path(i), row, 256)))
    print('decode_bgr (PIL full decode)       %.0f us' % t(lambda i: DS.decode_bgr(ds.img_path(i))))
    big = np.zeros((64, AJ.record_bytes(256)), np.uint8)
    print('file_to_record into 64 rotating rows %.0f us' % t(lambda i: AJ.file_to_record(ds.img_path(i), big[i % 64], 256)))
    sz = np.mean([os.path.getsize(ds.img_path(i)) for i in range(256)])
    print('mean file size %.0f bytes' % sz)
