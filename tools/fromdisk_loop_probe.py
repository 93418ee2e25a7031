"""Phase timers of the evaluation loop from files (apps/eval.py::evaluate_from_disk re-stated with timers): python tools/fromdisk_loop_probe.py [n] [workers] [source]"""
import itertools
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
from dir_amd.apps import jpeg as AJ  # noqa: E402
from dir_amd.engine import DirEngine, ForwardPipeline  # noqa: E402

if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    source = sys.argv[3] if len(sys.argv) > 3 else 'jpeg'
    bs = 256
    with open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json')) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    state = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
    with tempfile.TemporaryDirectory() as d:
        write_split(d, 256, seed=1)
        eng = DirEngine(state, dtype=torch.bfloat16)
        mano = DS.gt_layers_from_checkpoint(state)
        jreg = {s: EV.Jr(mano[s].J_regressor) for s in ('left', 'right')}
        eng.autotune(torch.randn(bs, 3, 256, 256, device='cuda'))
        dev = eng.device
        if source == 'u8':
            DS.write_u8_shards(d, 'test', shard_size=128, workers=8)
            ring = DS.ShardRing(d, 'test', bs, workers=workers, indices=[i % 256 for i in range(n)])
        else:
            ring = DS.DecodeRing(d, 'test', bs, workers=workers, indices=[i % 256 for i in range(n)], records=(source == 'jpeg'))
        m = EV.EvalMetrics(jreg, 0, True, 3)
        slots = [torch.zeros(bs, 256, 256, 3, device=dev, dtype=torch.uint8) for _ in range(2)]
        rec_dev = [torch.zeros(bs, ring.record_bytes, device=dev, dtype=torch.uint8) for _ in range(2)] if source == 'jpeg' else None
        rec_dec = [AJ.RecordDecoder(bs, ring.record_bytes, 256, dev) for _ in range(2)] if source == 'jpeg' else None
        pipe = ForwardPipeline(eng, slots, want_proj_feat=False)
        pending = [None, None]
        T = dict(ring=0.0, finish_wait=0.0, finish_metrics=0.0, copy_launch=0.0, copied_sync=0.0)

        def finish(slot):
            nn, annos = pending[slot]
            t0 = time.perf_counter()
            outs = pipe.wait(slot)
            t1 = time.perf_counter()
            res = [{k: (v[:nn] if torch.is_tensor(v) else v) for k, v in o.items()} for o in outs[:3]]
            gt = DS.gt_batch(mano, annos[:nn])
            m.update(res, (None,) * 2 + gt)
            T['finish_wait'] += t1 - t0
            T['finish_metrics'] += time.perf_counter() - t1
            pending[slot] = None
        t_all = time.perf_counter()
        it = iter(ring)
        k = 0
        while True:
            t0 = time.perf_counter()
            nxt = next(it, None)
            T['ring'] += time.perf_counter() - t0
            if nxt is None:
                break
            frames, annos, nn = nxt
            slot = k % 2
            if pending[slot] is not None:
                finish(slot)
            t0 = time.perf_counter()
            with torch.cuda.stream(pipe.streams[slot]):
                if source == 'jpeg':
                    rec_dev[slot].copy_(frames, non_blocking=True)
                    rec_dec[slot](rec_dev[slot], pipe.imgs[slot], nn)
                else:
                    pipe.imgs[slot].copy_(frames, non_blocking=True)
                annos_dev = annos.to(dev, non_blocking=True)
                copied = torch.cuda.Event()
                copied.record()
            pipe.launch(slot)
            pending[slot] = (nn, annos_dev)
            t1 = time.perf_counter()
            copied.synchronize()
            T['copy_launch'] += t1 - t0
            T['copied_sync'] += time.perf_counter() - t1
            k += 1
        for slot in ((k) % 2, (k - 1) % 2):
            if pending[slot] is not None:
                finish(slot)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t_all
        ring.close()
        print('%s workers %d: %d images in %.2f s = %.0f images/s; per batch (ms): %s' % (source, workers, n, dt, n / dt, {a: round(b / k * 1e3, 2) for a, b in T.items()}))
