"""Ring-only rates on the GPU box (no network): python tools/fromdisk_ring_probe.py [n] -- the decode ring alone, + DMA, + dir_jpeg_decode_records"""
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'helpers'))
from fake_split import write_split  # noqa: E402
from dir_amd.apps import dataset as DS  # noqa: E402
from dir_amd.apps import jpeg as AJ  # noqa: E402

if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    with tempfile.TemporaryDirectory() as d:
        write_split(d, 256, seed=1)
        idx = [i % 256 for i in range(n)]
        for rec in (True, False):
            for w in (8, 12, 16):
                for mode in ('ring', 'ring+dma+decode'):
                    if mode != 'ring' and not rec:
                        continue
                    ring = DS.DecodeRing(d, 'test', 256, workers=w, indices=idx, records=rec)
                    dev = [torch.zeros(256, ring.record_bytes if rec else 256 * 256 * 3, dtype=torch.uint8, device='cuda') for _ in range(2)]
                    out = torch.zeros(256, 256, 256, 3, dtype=torch.uint8, device='cuda')
                    dec = AJ.RecordDecoder(256, ring.record_bytes, 256) if rec else None
                    t0 = time.perf_counter()
                    seen = 0
                    for k, (fr, an, m) in enumerate(ring):
                        if mode != 'ring':
                            dev[k % 2].copy_(fr, non_blocking=True)
                            dec(dev[k % 2], out, m)
                            torch.cuda.current_stream().synchronize()
                        seen += m
                    dt = time.perf_counter() - t0
                    ring.close()
                    print('%-7s workers %2d %-16s %6.0f images/s' % ('records' if rec else 'pixels', w, mode, seen / dt), flush=True)
