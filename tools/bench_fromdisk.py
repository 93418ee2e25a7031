"""Sustained images/s of the evaluation loop FROM FILES (JPEG decode on the host cores -> uint8 frames -> two forwards in flight ->
GT MANO + metrics on the GPU), on a synthetic split in the reference's layout.  [NSLOT=3] [DTYPE=f16|bf16] python tools/bench_fromdisk.py [n_images] [bs] [workers]"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'helpers'))
from fake_split import write_split  # noqa: E402
from dir_amd import synth  # noqa: E402
from dir_amd.apps import dataset as DS  # noqa: E402
from dir_amd.apps import eval as EV  # noqa: E402
from dir_amd.engine import DirEngine  # noqa: E402

if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    workers = [int(w) for w in sys.argv[3].split(',')] if len(sys.argv) > 3 else [8, 16, 32, 64]
    with open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json')) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    state = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
    with tempfile.TemporaryDirectory() as d:
        t0 = time.perf_counter()
        write_split(d, 256, seed=1)
        print('wrote 256 synthetic 256x256 JPEGs + annotations in %.1f s (re-used cyclically as a %d-image split)' % (time.perf_counter() - t0, n))
        print('host cores usable by this process: %d (os.cpu_count() = %d)' % (len(os.sched_getaffinity(0)), os.cpu_count()))
        t0 = time.perf_counter()
        ds = DS.InterHandSplit(d)
        for i in range(64):
            ds.frame(i)
        print('one process decodes %.0f frames/s' % (64 / (time.perf_counter() - t0)))
        eng = DirEngine(state, dtype={'bf16': torch.bfloat16, 'f16': torch.float16}[os.environ.get('DTYPE', 'f16')])      # f16 storage: bench.py's headline mode
        mano = DS.gt_layers_from_checkpoint(state)
        jreg = {s: EV.Jr(mano[s].J_regressor) for s in ('left', 'right')}
        idx = [i % 256 for i in range(n)]
        eng.autotune(torch.randn(bs, 3, 256, 256, device='cuda'))
        # round 5: the workers decode the Huffman stream only (lib/libdir_jpeg.so), the GPU does the rest of cv.imread (dir_jpeg_decode_records)
        from dir_amd.apps import jpeg as AJ
        row = np.zeros(AJ.record_bytes(256), np.uint8)
        t0 = time.perf_counter()
        for i in range(256):
            AJ.file_to_record(ds.img_path(i), row, 256)
        print('one process entropy-decodes %.0f files/s into coefficient records (read + Huffman)' % (256 / (time.perf_counter() - t0)))
        # one short untimed pass first: the first scored batch of a process initialises the BLAS library behind the metric's J_regressor products
        # (a 0.2 s stall, found in the round-5 kernel trace: profiles/r05_fromdisk_sweep.txt) -- a per-process cost, not the loop's rate
        EV.evaluate_from_disk(eng, d, jreg, mano, bs=bs, workers=2, indices=idx[:4 * bs], source='jpeg', nslot=int(os.environ.get("NSLOT", "3")))
        for src in [s_ for s_ in os.environ.get('SOURCES', 'jpeg,jpeg-host').split(',') if s_ in ('jpeg', 'jpeg-host')]:
            for w in workers:
                m, rate = EV.evaluate_from_disk(eng, d, jreg, mano, bs=bs, workers=w, indices=idx, source=src, nslot=int(os.environ.get("NSLOT", "3")))
                print('%-9s workers %3d  bs %d: %d images in %.2f s = %.0f images/s from files' % (src, w, bs, rate['images'], rate['seconds'], rate['images_per_sec']))
        # the prepared uint8 split (dataset.write_u8_shards): the same loop without the JPEG decode
        t0 = time.perf_counter()
        DS.write_u8_shards(d, 'test', shard_size=128, workers=8)
        print('prepared uint8 shards of the 256 frames in %.1f s' % (time.perf_counter() - t0))
        for w in (2, 4, 8):
            m, rate = EV.evaluate_from_disk(eng, d, jreg, mano, bs=bs, workers=w, indices=idx, source='u8')
            print('u8 shards, %d copy threads  bs %d: %d images in %.2f s = %.0f images/s from files' % (w, bs, rate['images'], rate['seconds'], rate['images_per_sec']))
