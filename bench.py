#!/usr/bin/env python3
"""bench.py -- images/sec of DIR.forward (eval, 3 stage outputs) on synthetic 256x256 batches, BASELINE.json config 2
(batch 64 per GPU, ResNet-50 + init regression + 2 refinement stages, bf16 feature maps / fp32 token+MANO path).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

One step = one forward of the hot path over one batch already resident in HBM.  The forward is captured once in a
HIP graph and replayed.  Images are independent (eval-mode BN), so N GPUs shard the batch with NO data-path collective
(weak scaling: 64 images per GPU); the only collectives are the timing barrier and a MAX over ranks.
Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events around every launch of the dominant kernel
(the MFMA implicit-GEMM convolution) in an instrumented eager pass on the same stream; `cpu_baseline` times the numpy
oracle (oracle/, the CPU restatement of the reference) on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK = {'bf16': 2.5e15, 'f32': 157.3e12}        # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
ALG_GFLOP_PER_IMAGE = 36.80                     # BASELINE.md section 2 (reference, torch flop counter)
HBM_PEAK = 8.0e12                               # bytes/s, same guide


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64, help='images per GPU per step')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--inflight', type=int, default=2, help='forwards in flight per GPU (engine.ForwardPipeline: one captured graph + '
                    'stream + input batch per slot, steps alternate between them); 1 = one graph replayed back to back')
    ap.add_argument('--no-autotune', action='store_true', help='keep the library heuristic for every conv layer')
    ap.add_argument('--autotune-cache', default=None, help='JSON file: load the per-layer variants if it exists, else tune and save '
                    '(profiling runs use it to keep the exploration out of the trace)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=64, help='images timed on the CPU baseline')
    ap.add_argument('--cpu-threads', type=int, default=16)
    ap.add_argument('--dump-conv', action='store_true', help='print per-shape conv timings (stderr)')
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from dir_amd import dist as D
    from dir_amd import engine as E
    from dir_amd import synth

    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    rank, world, local = D.init_from_env('nccl', dev)          # nccl == RCCL on ROCm
    assert world == args.gpus or world == 1, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus

    with open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json')) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    sd_np = synth.synth_state_dict(shapes, 1234)
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}
    tdt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    eng = E.DirEngine(sd, dtype=tdt, device=dev)
    B = args.batch
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    img = torch.randn(B, 3, 256, 256, device=dev, generator=g)

    # ---- build the step (HIP graph of the whole forward)
    fwd = lambda: eng.forward(img)                # noqa: E731
    outs = fwd()                                  # eager once: allocator warm-up, lazy init
    torch.cuda.synchronize()
    if not args.no_autotune:                      # per-layer conv kernel variant for this batch size (bit-identical results)
        if args.autotune_cache and os.path.exists(args.autotune_cache):
            with open(args.autotune_cache) as f:
                eng.import_tuning(img, json.load(f))
        else:
            eng.autotune(img)
            if args.autotune_cache and rank == 0:
                with open(args.autotune_cache, 'w') as f:
                    json.dump(eng.export_tuning(B), f)
    serial_ms = None
    if args.no_graph:
        step = fwd
        args.inflight = 1
    elif args.inflight <= 1:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outs = fwd()
        step = graph.replay
    else:
        # steps alternate between `inflight` slots (own stream, graph, synthetic input batch; shared weights): step k's forward
        # overlaps step k-1's.  Every step is still one whole B-image forward; the timed region is closed by a device-wide sync.
        imgs = [img] + [torch.randn(B, 3, 256, 256, device=dev, generator=g) for _ in range(args.inflight - 1)]
        pipe = E.ForwardPipeline(eng, imgs)
        outs = pipe.outs[0]
        counter = [0]

        def step():
            pipe.launch(counter[0] % args.inflight)
            counter[0] += 1

        def one_slot():
            pipe.launch(0)
        for _ in range(3):
            one_slot()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            one_slot()
        torch.cuda.synchronize()
        serial_ms = (time.perf_counter() - t0) / 10 * 1e3      # same graphs, one forward at a time (reported beside `value`)

    def barrier():
        D.barrier(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = D.max_over_ranks(dt, dev)
    barrier()
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    finite = bool(torch.isfinite(outs[2]['pd_mesh_xyz_left']).all())
    # overlapped execution must not change results: every slot's outputs from the timed (overlapping) replays against the same
    # slot replayed alone (this is the check that exposed the packed-FP32 hazard)
    reproducible = None
    if not args.no_graph and args.inflight > 1:
        def snap(o):
            return [o[i][k].clone() for i in range(3) for k in ('pd_mesh_xyz_left', 'pd_mesh_xyz_right', 'pd_joint_uv_left', 'pd_offset')] + [o[3]['seg'].clone()]
        for s_ in range(args.inflight):
            pipe.launch(s_)
        torch.cuda.synchronize()
        overlapped = [snap(pipe.outs[s_]) for s_ in range(args.inflight)]
        reproducible = True
        for s_ in range(args.inflight):
            pipe.launch(s_)
            alone = snap(pipe.wait(s_))
            reproducible = reproducible and all(torch.equal(a, b) for a, b in zip(overlapped[s_], alone))

    # ---- roofline of the dominant kernel: HIP events around every conv launch, eager, same stream
    roof = None
    if rank == 0:
        tag = 'conv_igemm<%s,%s>' % (args.dtype, args.dtype)
        eng.overlap = False                       # per-kernel durations: no concurrent side-stream launches
        E.PROFILE = []
        eng.forward(img)
        torch.cuda.synchronize()
        E.PROFILE = []
        reps = 3
        for _ in range(reps):
            eng.forward(img)
        torch.cuda.synchronize()
        rec = [(r[0], r[1], r[2].elapsed_time(r[3])) for r in E.PROFILE]
        alg_bytes = [r[5] for r in E.PROFILE if r[0] == tag]
        if args.dump_conv:
            agg = {}
            for t_, f_, e0, e1, shp, by_, _ in E.PROFILE:
                a = agg.setdefault((t_, shp), [0, 0.0, 0.0, 0.0])
                a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += f_; a[3] += by_
            for (t_, shp), (n, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                sys.stderr.write('%-24s %-40s x%-3d %8.3f ms/step %8.1f TFLOP/s %8.1f GB/s\n' % (t_, shp, n // reps, ms / reps, fl / ms / 1e9, by / ms / 1e6))
        E_PROFILE = E.PROFILE
        E.PROFILE = None
        dom = [(f_, ms) for t_, f_, ms in rec if t_ == tag]
        n_launch = len(dom) // reps
        flops_per_launch = sum(f_ for f_, _ in dom) / len(dom)
        ms_per_launch = sum(ms for _, ms in dom) / len(dom)
        achieved = flops_per_launch / (ms_per_launch * 1e-3)
        conv_ms = sum(ms for _, _, ms in rec) / reps
        traffic, traffic_src = None, None
        import glob
        pm = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
        if pm and args.dtype == 'bf16' and B == 64:
            with open(pm[-1]) as f:
                traffic = round(json.load(f)['hbm_bytes_per_launch'])
            traffic_src = os.path.relpath(pm[-1], ROOT) + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)'
        # the conv family spans both regimes (K <= 512 1x1 layers stream, the 3x3 layers compute): price the aggregate
        # against both roofs and report the one it sits closer to as the binding one
        bytes_per_launch = sum(alg_bytes) / len(alg_bytes)
        hbm_rate = bytes_per_launch / (ms_per_launch * 1e-3)
        frac_mfma, frac_hbm = achieved / PEAK[args.dtype], hbm_rate / HBM_PEAK
        if frac_hbm >= frac_mfma:
            head = {'bound': 'hbm', 'kernel': tag, 'achieved': round(hbm_rate / 1e9, 1), 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                    'frac': round(frac_hbm, 4)}
        else:
            head = {'bound': 'mfma', 'kernel': tag, 'achieved': round(achieved / 1e12, 2), 'peak': PEAK[args.dtype] / 1e12,
                    'unit': 'TFLOP/s', 'frac': round(frac_mfma, 4)}
        # the same launches split by arithmetic intensity against the machine balance (peak FLOP/s : peak B/s): each class priced
        # against the roof that bounds it
        balance = PEAK[args.dtype] / HBM_PEAK
        cls = {'hbm': [0, 0.0, 0.0, 0.0], 'mfma': [0, 0.0, 0.0, 0.0]}
        for r, ms in zip([r for r in E_PROFILE if r[0] == tag], [ms for t_, _, ms in rec if t_ == tag]):
            c = cls['hbm' if r[1] / r[5] < balance else 'mfma']
            c[0] += 1; c[1] += ms; c[2] += r[1]; c[3] += r[5]
        by_class = {}
        for k, (n, ms, fl, by) in cls.items():
            if n:
                ach = (by if k == 'hbm' else fl) / (ms * 1e-3)
                pk = HBM_PEAK if k == 'hbm' else PEAK[args.dtype]
                by_class[k] = {'launches_per_step': n // reps, 'ms_per_step': round(ms / reps, 3),
                               'achieved': round(ach / (1e9 if k == 'hbm' else 1e12), 1), 'unit': 'GB/s' if k == 'hbm' else 'TFLOP/s',
                               'frac': round(ach / pk, 4)}
        roof = dict(head, by_class=by_class, traffic=traffic, traffic_source=traffic_src, frac_mfma=round(frac_mfma, 4), frac_hbm=round(frac_hbm, 4),
                    achieved_tflops=round(achieved / 1e12, 2), achieved_gbps=round(hbm_rate / 1e9, 1),
                    alg_bytes_per_launch=round(bytes_per_launch),
                    launches_per_step=n_launch, avg_launch_us=round(ms_per_launch * 1e3, 2),
                    alg_gflop_per_launch=round(flops_per_launch / 1e9, 3), all_conv_ms_per_step=round(conv_ms, 3),
                    whole_step_tflops=round(ALG_GFLOP_PER_IMAGE * 1e9 * B / (ms_per_step * 1e-3) / 1e12, 2))

    # ---- CPU baseline: the numpy oracle (CPU restatement of the reference) on the host cores (rank 0, single-GPU runs
    #      only), bounded sample.  OpenBLAS is pinned to the thread count that serves these GEMM sizes best on the box
    #      (measured: 8-16 threads 4.2 img/s, 32 threads 2.4, 64 threads 1.1 -- oversubscription), and `cores` reports it.
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.dir_forward import dir_forward
        from threadpoolctl import threadpool_limits
        nthreads = min(args.cpu_threads, os.cpu_count() or 1)
        n, chunk = args.cpu_sample, 8
        ximg = img[:min(n, B)].cpu().numpy()
        if n > B:
            ximg = np.concatenate([ximg] * ((n + B - 1) // B))[:n]
        with threadpool_limits(limits=nthreads):
            dir_forward(sd_np, ximg[:1])              # warm-up (BLAS thread pool, page faults)
            t0 = time.perf_counter()
            for i in range(0, n, chunk):
                dir_forward(sd_np, ximg[i:i + chunk])
            tc = time.perf_counter() - t0
        cpu = {'value': round(n / tc, 3), 'unit': 'images/sec', 'cores': int(nthreads), 'kind': 'port',
               'sample': '%d images in chunks of %d, fp32 forward of oracle/dir_forward.py (numpy + OpenBLAS, %d threads), '
                         '%.1f s' % (n, chunk, nthreads, tc)}

    if rank == 0:
        line = {'metric': 'images/sec at 256x256 bs=%d, 3 stage outputs (DIR.forward eval)' % B, 'value': round(value, 1),
                'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                'dtype': args.dtype, 'data': 'synthetic',
                'config': {'workload': 'BASELINE configs[1]: batch 64 synthetic 256x256 per GPU, ResNet-50 + init '
                                       'regression + 2 refinement stages (3 stage outputs), seg/dense/proj_feat heads',
                           'batch_per_gpu': B, 'graph': not args.no_graph, 'forwards_in_flight': args.inflight,
                           'ms_per_forward_one_in_flight': None if serial_ms is None else round(serial_ms, 3), 'weights': 'synthetic (dir_amd.synth seed 1234)',
                           'sharding': 'independent images per GPU, no data-path collective', 'outputs_finite': finite,
                           'overlapped_equals_one_at_a_time': reproducible},
                'roofline': roof, 'cpu_baseline': cpu}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
