#!/usr/bin/env python3
"""bench.py -- images/sec of DIR.forward (eval, 3 stage outputs) on synthetic 256x256 batches, BASELINE.json config 2
(batch 64 per GPU, ResNet-50 + init regression + 2 refinement stages, bf16 feature maps / fp32 token+MANO path).

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  Launched by the driver through torch.distributed.run, or -- when WORLD_SIZE is not in the
environment -- bench.py re-launches ITSELF through torch.distributed.run with N ranks (dir_amd.dist.spawn_ranks); a world size
that is not N is a hard error, never a silent 1-rank line.

One step = one forward of the hot path over one batch already resident in HBM.  The forward is captured once in a HIP graph and
replayed.  Images are independent (eval-mode BN), so N GPUs shard the batch with NO data-path collective (weak scaling: 64 images
per GPU); the only collectives are the timing barrier and a MAX over ranks.  The timed region (exactly K steps between barrier +
synchronize on both sides, MAX over ranks) is run `--repeats` times and the MEDIAN region is reported (all of them are listed).
Prints ONE JSON line (rank 0).

`roofline`: HIP events around every library call of an instrumented eager pass on the same stream, labelled with the kernel symbol
the library launched (dir_launch_log_get); `frac` etc. = the convolution family (the dominant kernels) as in round 1, `kernels` =
one entry per kernel symbol.  `cpu_baseline`: the numpy oracle (CPU restatement of the reference) and the same graph with its dense
operators on stock torch CPU kernels, on this box's host cores, on a bounded sample.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# One HIP hardware queue per forward in flight: the runtime multiplexes streams onto GPU_MAX_HW_QUEUES queues (default 4, one of them
# the null stream's), and two pipeline slots that share a queue do not overlap at all (measured: 4 slots on 4 queues 2.48 ms per step,
# on 8 queues 2.23 ms).  Must be set before the runtime initialises; an explicit setting of the user wins.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

PEAK = {'bf16': 2.5e15, 'f16': 2.5e15, 'f32': 157.3e12,         # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
        'f16x3': 2.5e15 / 3, 'f16': 2.5e15}                     # split precision: three f16 MFMAs (dense f16 peak = bf16's) per algorithmic product
ALG_GFLOP_PER_IMAGE = 36.80                     # BASELINE.md section 2 (reference, torch flop counter)
HBM_PEAK = 8.0e12                               # bytes/s, same guide


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--repeats', type=int, default=5, help='timed regions of --steps steps each; the median one is reported')
    ap.add_argument('--batch', type=int, default=64, help='images per GPU per step')
    ap.add_argument('--dtype', default='f16', choices=['bf16', 'f16', 'f32'], help="storage of feature maps and convolution weights: bf16 (BASELINE configs[1]), "
                    "f16 = the same data path and MFMA rate on IEEE f16 (11-bit significands: every stage inside 0.01 mm), f32 = exact parity mode")
    ap.add_argument('--weights', default='cond', choices=['cond', 'plain'], help="synthetic parameters: cond = trained-like (dir_amd.synth cond=True: activations O(1) in "
                    "every layer; the reference's own forward on them is golden G7c, whose two images ride in the timed batch as rows 5 and 63 -> `parity`), "
                    "plain = rounds 1-5's Kaiming-scale parameters (golden G7)")
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--inflight', type=int, default=4, help='forwards in flight per GPU (engine.ForwardPipeline: one captured graph + '
                    'stream + input batch per slot, steps alternate between them); 1 = one graph replayed back to back')
    ap.add_argument('--no-autotune', action='store_true', help='keep the library heuristic for every conv layer')
    ap.add_argument('--autotune-cache', default=None, help='JSON file: load the per-layer variants if it exists, else tune and save '
                    '(profiling runs use it to keep the exploration out of the trace)')
    ap.add_argument('--tuning', choices=['throughput', 'time'], default='throughput',
                    help='per-layer conv kernel table of the timed graphs: throughput = dir_amd/tuning/ (fewest joules per launch: the socket power '
                         'cap is what bounds several forwards in flight, DESIGN.md 9) when it matches this engine, else the live time-tuned choice')
    ap.add_argument('--force-table', action='store_true', help='load the throughput table even with one forward in flight / without graphs (profiling runs: the kernels of the timed graphs, one at a time)')
    ap.add_argument('--no-time-table-pass', action='store_true', help='skip the second per-launch pass with the time-tuned table (profiling runs: keeps the traces to the timed configuration)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-ceiling-probe', action='store_true', help='skip the in-run MFMA / HBM ceiling probe (2 s, outside the timed regions)')
    ap.add_argument('--no-power', action='store_true', help='skip the rocm-smi socket power probe (7 s, outside the timed regions)')
    ap.add_argument('--no-fp32-mode', action='store_true', help='skip the fp32 (exact-parity mode) sub-record')
    ap.add_argument('--no-pgcn', action='store_true', help='skip the P-GCN (SemGCN gather) sub-record')
    ap.add_argument('--no-other-half', action='store_true', help='skip the sub-record of the other 16-bit storage kind (f16 when --dtype bf16, bf16 when --dtype f16)')
    ap.add_argument('--no-config5', action='store_true', help='skip the BASELINE configs[4] (HRNet-W48, 32 per GPU) sub-record')
    ap.add_argument('--no-train', action='store_true', help='skip the training-step sub-record (batch 32, fp32)')
    ap.add_argument('--no-proj-feat-variant', action='store_true', help='skip the serving variant without the proj_feat output')
    ap.add_argument('--cpu-sample', type=int, default=32, help='images timed on the numpy CPU baseline')
    ap.add_argument('--cpu-threads', type=int, default=16)
    ap.add_argument('--detail-out', default='bench_detail.json', help='the full measurement record (the last stdout line is its compact form)')
    ap.add_argument('--dump-conv', action='store_true', help='print per-shape timings of every library call (stderr)')
    return ap.parse_args(argv)


def power_probe(step, sync, seconds, steps_per_burst):
    """Runs `step` back to back for `seconds` (outside every timed region) while a thread samples rocm-smi (dir_amd/power.py); the first
    second is dropped (the SMU's power reading is an average).  Returns the median socket power / shader clock, or None when rocm-smi
    gives nothing."""
    from dir_amd import power as P
    idle = P.smi_sample()
    if idle is None:
        return None
    smp = P.Sampler(skip=1.0, period=0.25).start()
    e0 = P.energy_joules()                       # the socket's energy accumulator (amdsmi): exact joules over the window
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(steps_per_burst):
            step()
        sync()
        n += steps_per_burst
    dt = time.perf_counter() - t0
    e1 = P.energy_joules()
    samples = smp.stop()
    if not samples:
        return None
    w = P.median(samples, 'w')
    ec = None if not (e0 and e1) else {'socket_w': round((e1[0] - e0[0]) / dt, 1), 'joules_per_step': round((e1[0] - e0[0]) / n, 3),
                                         'source': 'amdsmi energy accumulator over the whole window'}
    return {'energy_counter': ec, 'socket_w': w, 'cap_w': idle['cap'], 'frac_of_cap': round(w / idle['cap'], 3) if idle['cap'] else None,
            'sclk_mhz': P.median(samples, 'sclk'), 'w_before_probe': idle['w'], 'samples': len(samples),
            'ms_per_step_during_probe': round(dt / n * 1e3, 3), 'joules_per_step': round(w * dt / n, 3),
            'source': 'rocm-smi --showpower --showclocks, median of samples taken while the timed loop ran again for %.0f s' % seconds}


def timed_regions(step, steps, repeats, barrier, max_over_ranks, sync):
    """`repeats` regions of exactly `steps` steps, each bracketed by barrier + synchronize on both sides -> seconds per region (MAX
    over ranks)"""
    out = []
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        dt = time.perf_counter() - t0
        out.append(max_over_ranks(dt))
        barrier()
    return out


def kernel_table(records, reps, dtype):
    """one entry per kernel symbol: calls per step, average duration, algorithmic work, achieved rates and the roof it sits closer to"""
    agg = {}
    for r in records:
        a = agg.setdefault(r['kernels'] or r['api'], dict(calls=0, ms=0.0, flops=0.0, bytes=0.0, family=r.get('family', '?')))
        a['calls'] += 1
        a['ms'] += r['ms']
        a['flops'] += r.get('flops', 0.0)
        a['bytes'] += r.get('bytes', 0.0)
    rows = []
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]['ms']):
        sec = a['ms'] * 1e-3
        fm, fh = a['flops'] / sec / PEAK[dtype], a['bytes'] / sec / HBM_PEAK
        rows.append({'kernel': name, 'family': a['family'], 'calls_per_step': round(a['calls'] / reps, 2),
                     'avg_us': round(a['ms'] * 1e3 / a['calls'], 2), 'ms_per_step': round(a['ms'] / reps, 4),
                     'alg_gflop_per_call': round(a['flops'] / a['calls'] / 1e9, 3), 'alg_mb_per_call': round(a['bytes'] / a['calls'] / 1e6, 3),
                     'tflops': round(a['flops'] / sec / 1e12, 2), 'gbps': round(a['bytes'] / sec / 1e9, 1),
                     'bound': 'mfma' if fm >= fh else 'hbm', 'frac': round(max(fm, fh), 4)})
    return rows


def four_in_flight_profile(dtype):
    """Per-launch mean of the conv family WHILE four forwards are in flight, from the tracked rocprofv3 kernel trace of that configuration
    (tools/profile_four_in_flight.sh -> profiles/r*_kernel_stats_four_in_flight.txt: a trace cannot be taken inside the timed run).  The kernel
    names in that file are cut at 80 characters, so the family is matched by kernel name + storage kind; forwards = init_head_kernel launches."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_kernel_stats_four_in_flight.txt')))
    if not files or dtype not in ('bf16', 'f16'):
        return None
    fam = ('conv_igemm_kernel', 'conv_pipe_kernel', 'conv_patch_kernel', 'conv_big_kernel', 'bneck_chain_kernel', 'tail_chain_kernel', 'stream1x1_kernel', 'conv_as_kernel')
    calls, total, forwards = 0, 0.0, 0
    with open(files[-1]) as f:
        for ln in f:
            t = ln.split()
            if len(t) < 4 or not t[1].isdigit():
                continue
            is_f16 = 'f16s_t' in t[0]
            if 'init_head_kernel' in t[0] and is_f16 == (dtype == 'f16'):
                forwards += int(t[1])
            if any(k in t[0] for k in fam) and is_f16 == (dtype == 'f16'):
                calls += int(t[1])
                total += float(t[2])
    if not calls or not forwards:
        return None
    return {'source': os.path.relpath(files[-1], ROOT), 'forwards_traced': forwards, 'launches_per_forward': round(calls / forwards, 1),
            'avg_launch_us': round(total / calls, 2), 'conv_ms_per_forward': round(total / forwards / 1e3, 3),
            'note': 'kernel durations under contention (four graphs replaying on four streams): longer per launch than one at a time, overlapped in wall time'}


def main():
    args = parse_args()
    env_world = os.environ.get('WORLD_SIZE')
    if args.gpus > 1 and env_world is None:
        # bare `python bench.py --gpus N`: become the launcher of N ranks (one process per GPU over RCCL)
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus and os.environ.get('DIR_BENCH_BACKEND', 'nccl') == 'nccl':
            sys.stderr.write('bench.py: --gpus %d needs %d visible GPUs, this box has %d\n' % (args.gpus, args.gpus, have))
            sys.exit(2)
        from dir_amd import dist as D
        sys.exit(D.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    if int(env_world or 1) != args.gpus:          # checked BEFORE any rendezvous: a contradiction must fail, not hang or shrink
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%s: launch with torch.distributed.run --nproc-per-node %d (or with no '
                         'WORLD_SIZE in the environment, and bench.py launches the ranks itself)\n' % (args.gpus, env_world, args.gpus))
        sys.exit(2)

    import numpy as np
    import torch
    import torch.distributed as dist
    from dir_amd import _capi
    from dir_amd import dist as D
    from dir_amd import engine as E
    from dir_amd import synth

    if not torch.cuda.is_available():
        sys.stderr.write('bench.py: no GPU visible (the hot path has no CPU fallback)\n')
        sys.exit(2)
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # DIR_BENCH_BACKEND=gloo: the N > 1 control flow (rendezvous, barriers, max over ranks, rank 0's decisions and line) on a box with fewer GPUs
    # than ranks -- ranks then SHARE devices, which RCCL refuses and which makes `value` meaningless: a test of the code path
    # (tests/test_gpu_multi.py), never a measurement.  The line says which backend ran.
    backend = os.environ.get('DIR_BENCH_BACKEND', 'nccl')
    assert backend in ('nccl', 'gloo')
    if backend == 'gloo':
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    rank, world, local = D.init_from_env(backend, dev)          # nccl == RCCL on ROCm
    assert world == args.gpus
    world_observed = dist.get_world_size() if dist.is_initialized() else 1
    assert world_observed == world

    with open(os.path.join(ROOT, 'tests', 'golden', 'manifest_dir.json')) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    sd_np = synth.synth_state_dict(shapes, 1234, cond=args.weights == 'cond')
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}
    gold_name = 'g7c_dir' if args.weights == 'cond' else 'g7_dir'
    gold = dict(np.load(os.path.join(ROOT, 'tests', 'golden', gold_name + '.npz')))          # the reference's own DIR.forward on these parameters (oracle/gen_golden.py)
    GOLD_ROWS = (5, 63)
    tdt = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[args.dtype]
    half = args.dtype in ('bf16', 'f16')
    tbl_dtype = 'bf16' if half else args.dtype        # the shipped throughput table is keyed by layer shapes and kernel variants only: both 16-bit kinds share it
    eng = E.DirEngine(sd, dtype=tdt, device=dev)
    B = args.batch
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    img = torch.randn(B, 3, 256, 256, device=dev, generator=g)
    gold_rows = None
    if B > max(GOLD_ROWS):
        # the two golden images are rows of the TIMED batch (slot 0 of the pipeline), so the line carries its own parity
        # against the reference for the very graphs / kernel table / overlap it times (models/dir.py:513-540, apps/eval.py:167-172)
        gimg = torch.from_numpy(synth.synth_input('dir.img', (2, 3, 256, 256), 1234)).to(dev)
        img[GOLD_ROWS[0]], img[GOLD_ROWS[1]] = gimg[0], gimg[1]
        gold_rows = list(GOLD_ROWS)

    def parity_of(o, mode):
        """rows 5 / 63 of one forward's outputs against the reference golden: worst |xyz| over meshes and joints of every stage (m) and the mean
        per-joint position error per stage and hand (mm) -- the quantity BASELINE's 0.01 mm gate is about"""
        if gold_rows is None:
            return None
        worst, mp = 0.0, []
        for i in range(3):
            for side in ('left', 'right'):
                for k in ('pd_mesh_xyz_', 'pd_joint_xyz_'):
                    d = o[i][k + side][gold_rows].double().cpu().numpy() - gold['s%d.%s%s' % (i, k, side)]
                    worst = max(worst, float(np.abs(d).max()))
                    if k == 'pd_joint_xyz_':
                        mp.append(round(float(np.sqrt((d ** 2).sum(-1)).mean()) * 1e3, 5))
        return {'mode': mode, 'vs': gold_name.upper().replace('_DIR', '') + ' (reference DIR.forward, tests/golden/%s.npz), rows %d / %d of the timed batch' % (gold_name, GOLD_ROWS[0], GOLD_ROWS[1]),
                'max_abs_xyz_m': float('%.3e' % worst), 'mpjpe_mm_per_stage': mp, 'mpjpe_mm_max': max(mp),
                'meets_0p01mm_mpjpe': max(mp) < 0.01, 'meets_1e-4mm_positions': worst < 1e-7}

    def sync():
        torch.cuda.synchronize()

    def barrier():
        D.barrier(dev)

    def mx(v):
        return D.max_over_ranks(v, dev)

    # ---- build the step (HIP graph of the whole forward)
    fwd = lambda: eng.forward(img)                # noqa: E731
    outs = fwd()                                  # eager once: allocator warm-up, lazy init
    sync()
    if not args.no_autotune:                      # per-layer conv kernel variant for this batch size (bit-identical results)
        if args.autotune_cache and os.path.exists(args.autotune_cache):
            with open(args.autotune_cache) as f:
                eng.import_tuning(img, json.load(f))
        else:
            eng.autotune(img)
            if args.autotune_cache and rank == 0:
                with open(args.autotune_cache, 'w') as f:
                    json.dump(eng.export_tuning(B), f)
    serial_ms, serial_tp_ms, table_check = None, None, None
    pipe = None
    # Two kernel tables, the same bits out of both (checked below): the TIME-tuned one just made serves one forward at a time (latency); the timed
    # graphs take the THROUGHPUT table (dir_amd/tuning/, made by tools/energy_tune.py: fewest joules above idle per launch) when it matches.
    lat_graph, lat_outs, conv_tuning = None, None, 'time (live autotune)' if not args.no_autotune else 'library heuristic'
    if not args.no_graph and args.inflight > 1:
        lat_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(lat_graph):
            lat_outs = fwd()
        for _ in range(3):
            lat_graph.replay()
        sync()
        ser = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(10):
                lat_graph.replay()
            sync()
            ser.append((time.perf_counter() - t0) / 10 * 1e3)
        serial_ms = statistics.median(ser)      # one forward at a time, time-tuned kernels (reported beside `value`)
    t_time = None
    # (the throughput table is for several forwards in flight under captured graphs only: one forward alone runs ~15 % slower with it, so the
    # single-graph and eager paths keep the time-tuned choice -- ADVICE r3)
    if args.tuning == 'throughput' and not args.no_autotune and ((args.inflight > 1 and not args.no_graph) or args.force_table):
        t_time = eng.export_tuning(B)
        meta = eng.load_tuning_table(img, 'gfx950_%s_b%d_throughput' % (tbl_dtype, B))
        if meta is not None:
            conv_tuning = 'throughput table dir_amd/tuning/gfx950_%s_b%d_throughput.json (HEAD %s, %d layers differ from the time-tuned choice)' % (
                tbl_dtype, B, meta.get('head', '?'), meta.get('changed_vs_time_tuned', -1))
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
        table_check = None
        if t_time is not None and conv_tuning.startswith('throughput'):
            # The shipped table was measured on another box: check it here before it carries the timed regions.  Both tables' graphs on the same
            # streams, 3 x 20 steps each; the time-tuned graphs take over if the table does not pay on this machine (another power cap, ...).
            def quick(p_):
                k_ = [0]

                def st_():
                    p_.launch(k_[0] % args.inflight)
                    k_[0] += 1
                for _ in range(2 * args.inflight):
                    st_()
                sync()
                return statistics.median(timed_regions(st_, 20, 3, sync, float, sync)) / 20 * 1e3
            q_tp = quick(pipe)
            eng.import_tuning(img, t_time)
            pipe_t = E.ForwardPipeline(eng, imgs, streams=pipe.streams)
            q_t = quick(pipe_t)
            table_check = {'throughput_table_ms_per_step': round(q_tp, 3), 'time_tuned_ms_per_step': round(q_t, 3), 'steps': 60}
            # one decision for the whole job: rank 0's (every rank must time the same table under one `conv_tuning` label)
            take_time_tuned = mx(1.0 if (rank == 0 and q_t < 0.99 * q_tp) else 0.0) > 0.5
            if take_time_tuned:
                pipe, conv_tuning = pipe_t, 'time (live autotune): the shipped throughput table was slower on this machine (%.3f vs %.3f ms per step)' % (q_tp, q_t)
            else:
                del pipe_t
                eng.load_tuning_table(img, 'gfx950_%s_b%d_throughput' % (tbl_dtype, B))
        outs = pipe.outs[0]
        counter = [0]

        def step():
            pipe.launch(counter[0] % args.inflight)
            counter[0] += 1

        def one_slot():
            pipe.launch(0)
        for _ in range(3):
            one_slot()
        sync()
        ser = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(10):
                one_slot()
            sync()
            ser.append((time.perf_counter() - t0) / 10 * 1e3)
        serial_tp_ms = statistics.median(ser)   # the timed graphs, one forward at a time

    if pipe is not None:
        # settle: the table check above rebuilt and released graphs and ran 2 x 60 steps at the package power cap.  VERDICT r5 weak 9: the timed
        # regions must not inherit its thermal / clock state by accident -- run the timed loop itself for a fixed 0.4 s first (the state every
        # region then starts from is "this loop, running"), and report every region + their spread in the line.
        t_set = time.perf_counter()
        while time.perf_counter() - t_set < 0.4:
            for _ in range(3 * args.inflight):
                step()
            sync()
    for _ in range(args.warmup):
        step()
    regions = timed_regions(step, args.steps, max(1, args.repeats), barrier, mx, sync)
    dt = statistics.median(regions)
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    finite = bool(torch.isfinite(outs[2]['pd_mesh_xyz_left']).all())
    # overlapped execution must not change results: every slot's outputs from the timed (overlapping) replays against the same
    # slot replayed alone (this is the check that exposed the packed-FP32 hazard)
    reproducible = None
    if pipe is not None:
        def snap(o):
            return [o[i][k].clone() for i in range(3) for k in ('pd_mesh_xyz_left', 'pd_mesh_xyz_right', 'pd_joint_uv_left', 'pd_offset')] + [o[3]['seg'].clone()]
        for s_ in range(args.inflight):
            pipe.launch(s_)
        sync()
        overlapped = [snap(pipe.outs[s_]) for s_ in range(args.inflight)]
        reproducible = True
        for s_ in range(args.inflight):
            pipe.launch(s_)
            alone = snap(pipe.wait(s_))
            reproducible = reproducible and all(torch.equal(a, b) for a, b in zip(overlapped[s_], alone))
    # the headline's own parity: slot 0 (it holds `img`, golden rows included) out of an OVERLAPPED round of the timed graphs
    headline_parity = None
    if rank == 0:
        if pipe is not None:
            for s_ in range(args.inflight):
                pipe.launch(s_)
            sync()
            headline_parity = parity_of(pipe.outs[0], args.dtype + (' storage' if half else ''))
        else:
            step()
            sync()
            headline_parity = parity_of(outs, args.dtype + (' storage' if half else ''))
    tunings_equal = None
    if pipe is not None and lat_graph is not None:        # slot 0 and the latency graph read the same images: both kernel tables, the same bits
        lat_graph.replay()
        pipe.launch(0)
        a_, b_ = snap(lat_outs), snap(pipe.wait(0))
        sync()
        tunings_equal = all(torch.equal(x_, y_) for x_, y_ in zip(a_, b_))

    # ---- socket power while the same loop runs (after the timed regions; four forwards in flight sit at the package power cap: DESIGN.md 9)
    power = None
    if rank == 0 and world == 1 and not args.no_power:
        power = power_probe(step, sync, 4.0, 50)
        if power is not None and pipe is not None:
            p1 = power_probe(lat_graph.replay if lat_graph is not None else one_slot, sync, 3.0, 20)
            power['one_in_flight'] = None if p1 is None else {k: p1[k] for k in ('socket_w', 'sclk_mhz', 'ms_per_step_during_probe', 'joules_per_step', 'energy_counter')}

    # ---- serving variant: the same step without the proj_feat output (335 MB of fp32 per step that apps/eval.py:170-172 never
    #      reads).  Reported beside the headline, never as `value`.
    no_pf = None
    if rank == 0 and world == 1 and pipe is not None and not args.no_proj_feat_variant:
        pipe2 = E.ForwardPipeline(eng, pipe.imgs, want_proj_feat=False, streams=pipe.streams)      # same hardware queues as the headline pipeline
        c2 = [0]

        def step2():
            pipe2.launch(c2[0] % args.inflight)
            c2[0] += 1
        for _ in range(args.warmup):
            step2()
        r2 = timed_regions(step2, args.steps, 3, sync, float, sync)
        d2 = statistics.median(r2)
        r3 = timed_regions(step, args.steps, 3, sync, float, sync)          # the headline pipeline again, right after: same clocks
        d3 = statistics.median(r3)
        no_pf = {'images_per_sec': round(B * args.steps / d2, 1), 'ms_per_step': round(d2 / args.steps * 1e3, 3),
                 'with_proj_feat_measured_right_after_ms_per_step': round(d3 / args.steps * 1e3, 3)}
        del pipe2

    # ---- roofline: HIP events around every library call, eager, same stream
    def live_roofline(eng, img, dtype_key, ms_per_step, with_traffic=True):
        eng.overlap = False                       # per-kernel durations: no concurrent side-stream launches
        reps = 3
        _capi.PROFILE = []
        eng.forward(img)
        sync()
        _capi.PROFILE = []
        for _ in range(reps):
            eng.forward(img)
        sync()
        recs = _capi.PROFILE
        _capi.PROFILE = None
        for r in recs:
            r['ms'] = r['e0'].elapsed_time(r['e1'])
        if args.dump_conv:
            agg = {}
            for r in recs:
                a = agg.setdefault((r['kernels'], r.get('shape', r['api'])), [0, 0.0, 0.0, 0.0])
                a[0] += 1; a[1] += r['ms']; a[2] += r.get('flops', 0.0); a[3] += r.get('bytes', 0.0)
            for (t_, shp), (n, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                sys.stderr.write('%-24s %-60s x%-3d %8.3f ms/step %8.1f TFLOP/s %8.1f GB/s\n' % (t_, shp, n // reps, ms / reps, fl / ms / 1e9, by / ms / 1e6))
        conv = [r for r in recs if r.get('family') == 'conv']
        n_launch = len(conv) // reps
        flops_per_launch = sum(r['flops'] for r in conv) / len(conv)
        bytes_per_launch = sum(r['bytes'] for r in conv) / len(conv)
        ms_per_launch = sum(r['ms'] for r in conv) / len(conv)
        achieved = flops_per_launch / (ms_per_launch * 1e-3)
        hbm_rate = bytes_per_launch / (ms_per_launch * 1e-3)
        conv_ms = sum(r['ms'] for r in conv) / reps
        all_ms = sum(r['ms'] for r in recs) / reps
        traffic, traffic_src = None, None
        import glob
        pm = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
        if pm and dtype_key in ('bf16', 'f16') and B == 64 and with_traffic:
            with open(pm[-1]) as f:
                pj = json.load(f)
            traffic = round(pj['hbm_bytes_per_launch'])
            traffic_src = (os.path.relpath(pm[-1], ROOT) + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command at HEAD %s; a tracked '
                           'file, not re-measured inside this run: counters need their own rocprofv3 passes)' % pj.get('head', 'unrecorded'))
        # the conv family spans both regimes (K <= 512 1x1 layers stream, the 3x3 layers compute): price the aggregate
        # against both roofs and report the one it sits closer to as the binding one
        frac_mfma, frac_hbm = achieved / PEAK[dtype_key], hbm_rate / HBM_PEAK
        fam = 'conv family: ' + ' + '.join(sorted({k for r in conv for k in r['kernels'].split(',')}))
        if frac_hbm >= frac_mfma:
            head = {'bound': 'hbm', 'kernel': fam, 'achieved': round(hbm_rate / 1e9, 1), 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                    'frac': round(frac_hbm, 4)}
        else:
            head = {'bound': 'mfma', 'kernel': fam, 'achieved': round(achieved / 1e12, 2), 'peak': PEAK[dtype_key] / 1e12,
                    'unit': 'TFLOP/s', 'frac': round(frac_mfma, 4)}
        # the same launches split by arithmetic intensity against the machine balance (peak FLOP/s : peak B/s): each class priced
        # against the roof that bounds it
        balance = PEAK[dtype_key] / HBM_PEAK
        cls = {'hbm': [0, 0.0, 0.0, 0.0], 'mfma': [0, 0.0, 0.0, 0.0]}
        for r in conv:
            c = cls['hbm' if r['flops'] / r['bytes'] < balance else 'mfma']
            c[0] += 1; c[1] += r['ms']; c[2] += r['flops']; c[3] += r['bytes']
        by_class = {}
        for k, (n, ms, fl, by) in cls.items():
            if n:
                ach = (by if k == 'hbm' else fl) / (ms * 1e-3)
                pk = HBM_PEAK if k == 'hbm' else PEAK[dtype_key]
                by_class[k] = {'launches_per_step': n // reps, 'ms_per_step': round(ms / reps, 3),
                               'achieved': round(ach / (1e9 if k == 'hbm' else 1e12), 1), 'unit': 'GB/s' if k == 'hbm' else 'TFLOP/s',
                               'frac': round(ach / pk, 4)}
        executed = sum(r.get('flops', 0.0) for r in recs) / reps
        return dict(head, by_class=by_class, traffic=traffic, traffic_source=traffic_src, frac_mfma=round(frac_mfma, 4), frac_hbm=round(frac_hbm, 4),
                    achieved_tflops=round(achieved / 1e12, 2), achieved_gbps=round(hbm_rate / 1e9, 1),
                    alg_bytes_per_launch=round(bytes_per_launch),
                    launches_per_step=n_launch, avg_launch_us=round(ms_per_launch * 1e3, 2),
                    alg_gflop_per_launch=round(flops_per_launch / 1e9, 3), all_conv_ms_per_step=round(conv_ms, 3),
                    all_kernels_ms_per_step=round(all_ms, 3), library_calls_per_step=len(recs) // reps,
                    # FLOPs the kernels of one step actually execute (the bf16 mode factorises the 15.1 GFLOP/image fusion conv to
                    # K = 720) over the step time, and the reference's 36.8 GFLOP/image over the same time (an effective rate: how
                    # fast the reference's arithmetic would have to run to keep up -- not a utilisation)
                    executed_tflops=round(executed / (ms_per_step * 1e-3) / 1e12, 2),
                    effective_reference_tflops=round(ALG_GFLOP_PER_IMAGE * 1e9 * B / (ms_per_step * 1e-3) / 1e12, 2),
                    kernels=kernel_table(recs, reps, dtype_key))


    roof = live_roofline(eng, img, args.dtype, ms_per_step) if rank == 0 else None
    if roof is not None and serial_ms is not None:
        # `all_kernels_ms_per_step` is the SUM of HIP-event brackets around every library call of an EAGER forward (each bracket =
        # the call's kernel(s) + its own dispatch gap on an otherwise idle stream); `ms_per_forward_one_in_flight` is ONE graph replay of the same
        # launches (no host in the loop, dispatches back to back).  Their difference over the call count is the eager bracket's overhead per call.
        roof['eager_bracket_overhead_us_per_call'] = round((roof['all_kernels_ms_per_step'] - serial_ms) * 1e3 / max(roof['library_calls_per_step'], 1), 2)
        roof['reconcile'] = ('all_kernels_ms_per_step %.3f (eager, %d event-bracketed library calls, time-tuned or throughput table as timed) = ms_per_forward_one_in_flight '
                             '%.3f (one HIP-graph replay, time-tuned table) + %.2f us per call of dispatch gap / event granularity (+ the table difference when the throughput '
                             'table is loaded: see time_tuned_table.all_conv_ms_per_step)' % (roof['all_kernels_ms_per_step'], roof['library_calls_per_step'], serial_ms,
                                                                                               roof['eager_bracket_overhead_us_per_call']))
    if roof is not None and half and not args.no_ceiling_probe:
        # What THIS board sustains, measured in this run (dir_probe_launch: VERDICT r3 item 9 -- the constants quoted here in round 3 came from
        # another box): a bf16 MFMA loop on pseudo-random operands held in registers (the guide's 2.5 PFLOP/s is reached on all-zero operands
        # only: on real data the board's power / current limits pull the clock) and a streaming read of 1 GiB.  ~1 s each, after the timed
        # regions.  Reported beside `peak`, never instead of it.
        def probe(mode, buf, nbytes, iters, seconds):
            sp = torch.cuda.current_stream().cuda_stream
            L = _capi.lib()
            per = L.dir_probe_launch(mode, _capi.ptr(buf), nbytes, iters, sp)
            _capi.check(0 if per > 0 else int(per), 'dir_probe_launch')
            sync()
            t0, work = time.perf_counter(), 0
            while time.perf_counter() - t0 < seconds:
                for _ in range(4):
                    work += L.dir_probe_launch(mode, _capi.ptr(buf), nbytes, iters, sp)
                sync()
            return work / (time.perf_counter() - t0)
        big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        big.fill_(1)
        ce = {'mfma_bf16_random_operands_in_registers_tflops': round(probe(0, big, 64, 20000, 1.0) / 1e12, 1),
              'mfma_f16_random_operands_in_registers_tflops': round(probe(3, big, 64, 20000, 1.0) / 1e12, 1),
              'hbm_read_1gib_tbps': round(probe(1, big, 1 << 30, 0, 1.0) / 1e12, 3),
              # a float4 COPY of 512 MiB -> 512 MiB (bytes read + written): what a streaming kernel does; the read-only loop under-reports the ceiling
              # (5.1 TB/s, below bone_vis_kernel's own 6.3 TB/s)
              'hbm_copy_1gib_tbps': round(probe(2, big, 1 << 30, 0, 1.0) / 1e12, 3),
              # round 6: the two on-chip roofs of a small-pixel-tile convolution -- every CU streaming the SAME 1.2 MB (one 3x3 256 -> 256 layer's weights)
              # out of the L2s, and ds_read_b128 from LDS (DESIGN.md "balance": 64 FLOP per weight byte x this rate bounds the 64-pixel tiles)
              'l2_same_stream_all_cus_tbps': round(probe(4, big, 1179648, 64, 0.5) / 1e12, 2),
              'lds_read_all_cus_tbps': round(probe(5, big, 64, 2000, 0.5) / 1e12, 1),
              'source': 'dir_probe_launch in this run, ~1 s per loop, after the timed regions'}
        del big
        if 'mfma' in roof.get('by_class', {}):
            ce['by_class_mfma_frac_of_measured'] = round(roof['by_class']['mfma']['achieved'] / ce['mfma_%s_random_operands_in_registers_tflops' % args.dtype], 3)
        if 'hbm' in roof.get('by_class', {}):
            ce['by_class_hbm_frac_of_measured'] = round(roof['by_class']['hbm']['achieved'] / 1e3 / max(ce['hbm_read_1gib_tbps'], ce['hbm_copy_1gib_tbps']), 3)
        roof['measured_ceilings'] = ce
    if roof is not None and t_time is not None and conv_tuning.startswith('throughput') and not args.no_time_table_pass:
        # the same pass with the TIME-tuned table: the throughput table trades per-kernel duration (one forward alone) for joules, so its launches
        # look slower one at a time than the kernels can run
        eng.import_tuning(img, t_time)
        dump, args.dump_conv = args.dump_conv, False
        rt = live_roofline(eng, img, args.dtype, ms_per_step, with_traffic=False)
        args.dump_conv = dump
        roof['time_tuned_table'] = {k: rt[k] for k in ('frac_mfma', 'frac_hbm', 'by_class', 'all_conv_ms_per_step', 'avg_launch_us')}
        fl_step = roof['alg_gflop_per_launch'] * 1e9 * roof['launches_per_step']
        roof['overlapped'] = {'achieved': round(fl_step / (ms_per_step * 1e-3) / 1e12, 2), 'unit': 'TFLOP/s', 'frac': round(fl_step / (ms_per_step * 1e-3) / PEAK[args.dtype], 4),
                              'note': "the family's algorithmic FLOPs of one step over the whole timed step (%d forwards in flight: its launches overlap "
                                      "other forwards' kernels, so per-launch durations do not add up to the step)" % args.inflight}
        fif = four_in_flight_profile(args.dtype)
        if fif is not None:
            roof['overlapped']['four_in_flight_profile'] = fif
        roof['note_tables'] = ('kernel durations above: the table of the timed graphs (throughput); time_tuned_table: the same launches with '
                               'the per-layer fastest variant (what one forward at a time runs)')
        eng.load_tuning_table(img, 'gfx950_%s_b%d_throughput' % (tbl_dtype, B))

    # ---- the OTHER 16-bit storage kind (bf16 <-> f16), same kernels' twin instantiations, same kernel table, same streams, measured right after
    #      the headline and again beside it (alternating regions: same clocks): VERDICT r4 item 2 -- "an fp16-storage throughput mode that meets
    #      0.01 mm at every stage at bf16 speed" (tests/test_gpu_dir.py::test_engine_vs_reference_golden_trained_like_weights[f16s])
    other_half = None
    if rank == 0 and world == 1 and half and pipe is not None and not args.no_other_half:
        odt, oname = (torch.float16, 'f16') if args.dtype == 'bf16' else (torch.bfloat16, 'bf16')
        engo = E.DirEngine(sd, dtype=odt, device=dev)
        engo.forward(img)
        sync()
        if not args.no_autotune:
            engo.import_tuning(img, eng.export_tuning(B))          # the headline's kernel table (layer shapes are identical)
        pipeo = E.ForwardPipeline(engo, pipe.imgs, streams=pipe.streams)
        co = [0]

        def stepo():
            pipeo.launch(co[0] % args.inflight)
            co[0] += 1
        for _ in range(args.warmup + 2 * args.inflight):
            stepo()
        sync()
        ro, rh = [], []
        for _ in range(3):                                         # alternate: other kind, headline kind
            ro += timed_regions(stepo, args.steps, 1, sync, float, sync)
            rh += timed_regions(step, args.steps, 1, sync, float, sync)
        do, dh = statistics.median(ro), statistics.median(rh)
        for s_ in range(args.inflight):
            pipeo.launch(s_)
        sync()
        par_o = parity_of(pipeo.outs[0], oname + ' storage')
        other_half = {'parity': par_o, 'dtype': oname, 'images_per_sec': round(B * args.steps / do, 1), 'ms_per_step': round(do / args.steps * 1e3, 3),
                      'forwards_in_flight': args.inflight, 'headline_dtype_alternating_ms_per_step': round(dh / args.steps * 1e3, 3),
                      'steps': args.steps, 'regions': 3,
                      'note': 'DirEngine(dtype=%s): feature maps and convolution weights stored as %s, the same data path / kernel table / streams as the '
                              'headline; regions alternate with the headline pipeline' % (str(odt)[6:], oname)}
        if roof is not None:
            # the same per-launch pass as `roofline` (one forward in flight, the timed kernel table), so the two storage kinds compare like for like
            ro_ = live_roofline(engo, img, oname, do / args.steps * 1e3, with_traffic=False)
            other_half['roofline'] = {k: ro_[k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'frac_mfma', 'frac_hbm', 'avg_launch_us', 'all_conv_ms_per_step') if k in ro_}
        del pipeo, engo

    # ---- the SemGCN gather path alone (north_star: ">= 60 % of HBM peak on the SemGCN gather at batch 64"): the P-GCN stack of both
    #      hands (4 PGraphConv layers + mix = 5 dependent launches, dir_pgcn_stack_forward_pair) timed with HIP events at B = 64 and at the batch where
    #      it saturates, algorithmic bytes = weights once + every layer's activations in and out, beside the counted PMC bytes of the
    #      tracked sweep (profiles/r03_pgcn_batch_sweep_pmc.txt: FETCH_SIZE x 2 + WRITE_SIZE per launch)
    pgcn = None
    if rank == 0 and world == 1 and half and not args.no_pgcn:
        import ctypes as C
        st3 = eng.stage3
        Lb = _capi.lib()
        pgcn = {'target': '>= 0.60 of 8 TB/s at batch 64 (BASELINE.json north_star)'}
        for Bp in (64, 1024):
            x0p = torch.randn(2, Bp, 21, 128, device=dev)
            gpp = torch.randn(2, Bp, 21, 128, device=dev)
            tokp = torch.empty(Bp, 42, 128, device=dev)
            scp = torch.empty(4, Bp, 21, 256, device=dev)
            sp_ = torch.cuda.current_stream().cuda_stream

            def call():
                _capi.check(Lb.dir_pgcn_stack_forward_pair(st3.gcn[0], st3.gcn[1], 4, _capi.ptr(x0p), _capi.ptr(gpp), _capi.ptr(tokp), _capi.ptr(scp), Bp, C.c_void_p(sp_)), 'pgcn')
            for _ in range(5):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                call()
            e1.record()
            sync()
            us = e0.elapsed_time(e1) / 50 * 1e3
            alg = 4 * 2 * 2 * 21 * 128 * 128 * 2 + 4 * 2 * 2 * Bp * 21 * 128 * 4
            pgcn['B=%d' % Bp] = {'us_per_stack_pair': round(us, 1), 'alg_mb': round(alg / 1e6, 1), 'alg_gbps': round(alg / us / 1e3, 1),
                                 'frac_of_hbm_peak': round(alg / (us * 1e-6) / HBM_PEAK, 4)}
        pgcn['pmc_counted'] = ('profiles/r03_pgcn_batch_sweep_pmc.txt (same kernels: tokens.hip pgcn_node / pgcn_mix unchanged since): B=64 40 MB counted per stack in 34.5 us = '
                               '1.2 TB/s (0.15); B=1024 479 MB in 114.6 us = 4.2 TB/s (0.52); B=4096 1976 MB in 494 us = 4.0 TB/s (0.50) -- counted bytes exceed the '
                               'algorithmic ones 2.6x at large batch (the W0 | W1 halves travel through a 256-wide scratch tensor between node and mix launches)')
        pgcn['latency_bound'] = ('at B = 64 the stack is 5 DEPENDENT launches that move <= 10 MB each (1.2 us at 8 TB/s): each costs one launch boundary + one gather -> '
                                 'barrier -> weight fragment -> MFMA -> store chain, ~5 us measured floor per layer (one-launch rebuild with flag hand-offs: 29.9 us, DESIGN.md 10); '
                                 '0.60 of peak would be 4.6 us for the whole stack -- below ONE launch; the rate the kernels sustain appears at B >= 1024')

    # ---- fp32 exact-parity mode (the mode that meets the 1e-4 mm budget, tests/test_gpu_dir.py): one graph, a few steps
    fp32 = None
    if rank == 0 and world == 1 and half and not args.no_fp32_mode and not args.no_graph:
        del pipe
        eng32 = E.DirEngine(sd, dtype=torch.float32, device=dev)
        eng32.forward(img)
        sync()
        g32 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g32):
            o32 = eng32.forward(img)
        for _ in range(2):
            g32.replay()
        r32 = timed_regions(g32.replay, 5, 3, sync, float, sync)
        d32 = statistics.median(r32)
        fp32 = {'parity': parity_of(o32, 'f32'), 'images_per_sec': round(B * 5 / d32, 1), 'ms_per_step': round(d32 / 5 * 1e3, 3), 'steps': 5, 'regions': 3,
                'note': 'DirEngine(dtype=float32): exact fp32 MFMA everywhere, the mode the 1e-4 mm parity tests run; one forward in flight'}
        del g32, eng32, o32

    # ---- split-precision parity mode (DirEngine(dtype=float32, arith='f16x3'): fp32 feature maps and token path, convolutions on the
    #      f16 matrix cores with hi / lo operands, 3 products per multiply): the 1e-4 mm tests pass in it as in the exact-fp32 mode
    #      (tests/test_gpu_dir.py::test_engine_fp32_vs_reference_golden[f16x3]); first-class sub-record with its own roofline
    def arith_mode(arith, note):
        engx = E.DirEngine(sd, dtype=torch.float32, device=dev, arith=arith)
        engx.calibrate(img)
        engx.forward(img)
        sync()
        if not args.no_autotune:
            engx.autotune(img)
        # one forward at a time: a graph with the time-tuned table; the overlapped graphs take the throughput table when one is shipped for this mode
        gx = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gx):
            engx.forward(img)
        for _ in range(3):
            gx.replay()
        sync()
        r1 = timed_regions(gx.replay, 5, 3, sync, float, sync)
        del gx
        tuning_x = 'time (live autotune)'
        if args.tuning == 'throughput' and not args.no_autotune and engx.load_tuning_table(img, 'gfx950_%s_b%d_throughput' % (arith, B)) is not None:
            tuning_x = 'throughput table dir_amd/tuning/gfx950_%s_b%d_throughput.json' % (arith, B)
        nslot = max(1, min(args.inflight, 4))
        imgs_x = [img] + [torch.randn(B, 3, 256, 256, device=dev, generator=g) for _ in range(nslot - 1)]
        pipex = E.ForwardPipeline(engx, imgs_x)
        cx = [0]

        def stepx():
            pipex.launch(cx[0] % nslot)
            cx[0] += 1
        for _ in range(3):
            pipex.launch(0)
        sync()
        for _ in range(args.warmup):
            stepx()
        rx = timed_regions(stepx, 10, 3, sync, float, sync)
        for s_ in range(nslot):
            pipex.launch(s_)
        sync()
        par_x = parity_of(pipex.outs[0], arith)             # slot 0 holds `img`: the golden rows, out of an overlapped round
        dx, d1 = statistics.median(rx), statistics.median(r1)
        rec = {'images_per_sec': round(B * 10 / dx, 1), 'ms_per_step': round(dx / 10 * 1e3, 3), 'steps': 10, 'regions': 3,
               'forwards_in_flight': nslot, 'ms_per_forward_one_in_flight': round(d1 / 5 * 1e3, 3), 'dtype': arith, 'conv_tuning': tuning_x,
               'speedup_over_fp32_mode': None if fp32 is None else round(fp32['ms_per_step'] / (dx / 10 * 1e3), 2), 'note': note, 'parity': par_x,
               'roofline': live_roofline(engx, img, arith if arith in PEAK else 'bf16', dx / 10 * 1e3, with_traffic=False)}
        del pipex, engx
        return rec

    parity, f16m = None, None
    if rank == 0 and world == 1 and half and not args.no_fp32_mode and not args.no_graph:
        parity = arith_mode('f16x3', "DirEngine(dtype=float32, arith='f16x3'): fp32 feature maps / token path, every convolution product as three f16 "
                            'MFMAs (hi*hi + lo*hi + hi*lo, fp32 accumulate); meets the 1e-4 mm budget like fp32_mode (8.0e-8 m vs the reference '
                            'golden); roofline priced against the dense f16 peak / 3')
        f16m = arith_mode('f16', "DirEngine(dtype=float32, arith='f16'): the fp16 MFMA path of BASELINE config 5 -- fp32 feature maps, convolution operands "
                          'rounded to f16 (one MFMA per product, fp32 accumulate); all three stages within 0.01 mm of the reference on trained-like '
                          'weights (init 0.007 mm, refined 0.0003 - 0.001 mm; bf16: 0.05 / 0.005); roofline priced against the dense f16 peak')

    # ---- BASELINE configs[4] on one GPU: HRNet-W48 + init regression + 4 refinement stages ("5 refinement iters"), 32 images per GPU (batch 256
    #      over 8), in bf16 and in the arithmetic the config names (fp16 MFMA path: fp32 feature maps, f16 operands); graph + live autotune +
    #      the same forwards in flight as the headline.  No reference counterpart: parity is pinned to the build's oracle.
    cfg5 = None
    if rank == 0 and world == 1 and half and not args.no_config5 and not args.no_graph:
        from dir_amd.models.dir import DIR as _DIR
        torch.cuda.empty_cache()
        net5 = _DIR(21, 'x', 0, backbone='hrnet_w48', extra_stages=2)
        shapes5 = {k: tuple(v.shape) for k, v in net5.state_dict().items()}
        del net5
        sd5 = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes5, 1234, cond=True).items()}
        B5, nslot = 32, max(1, min(args.inflight, 4))
        imgs5 = [torch.randn(B5, 3, 256, 256, device=dev, generator=g) for _ in range(nslot)]
        cfg5 = {'batch_per_gpu': B5, 'forwards_in_flight': nslot, 'workload': 'BASELINE configs[4] per GPU: HRNet-W48 + init regression + 4 refinement stages, 32 images'}
        # bf16 storage | f16 STORAGE (round 5: the bf16 data path on IEEE f16 maps and weights, fp16 MFMAs -- the 'fp16' configs[4] names, at bf16 speed) |
        # fp32 storage with one f16 MFMA per product (rounds 3-4's reading of 'fp16': the slow, scale-calibrated path)
        for tag, dt5, ar5 in (('bf16', torch.bfloat16, None), ('f16s', torch.float16, None), ('fp16', torch.float32, 'f16')):
            e5 = E.DirEngine(sd5, dtype=dt5, device=dev, arith=ar5)
            e5.calibrate(imgs5[0])
            e5.forward(imgs5[0])
            sync()
            if not args.no_autotune:
                e5.autotune(imgs5[0], reps=1)
            if tag == 'bf16':          # share of the convolutions' MFMA work spent on zero-padded channels (widths 48 / 96 run as 64 / 128)
                e5.overlap = False
                _capi.PROFILE = []
                e5.forward(imgs5[0])
                sync()
                recs5, _capi.PROFILE = _capi.PROFILE, None
                ex5 = sum(r.get('flops', 0.0) for r in recs5 if r.get('family') == 'conv')
                re5 = sum(r.get('flops_real', r.get('flops', 0.0)) for r in recs5 if r.get('family') == 'conv')
                cfg5['pad_waste'] = round(1.0 - re5 / ex5, 3)
                cfg5['launches_per_step'] = len(recs5)
                cfg5['conv_gflop_per_image_executed'] = round(ex5 / B5 / 1e9, 1)
            p5 = E.ForwardPipeline(e5, imgs5)
            c5 = [0]

            def step5():
                p5.launch(c5[0] % nslot)
                c5[0] += 1
            for _ in range(2 * nslot):
                step5()
            sync()
            r5 = timed_regions(step5, 10, 3, sync, float, sync)
            d5 = statistics.median(r5)
            for _ in range(2):
                p5.launch(0)
            sync()
            r51 = timed_regions(lambda: p5.launch(0), 5, 3, sync, float, sync)
            pre5 = {'bf16': '', 'f16s': 'f16_storage_', 'fp16': 'fp16_'}[tag]
            cfg5[pre5 + 'images_per_sec'] = round(B5 * 10 / d5, 1)
            cfg5[pre5 + 'ms_per_step'] = round(d5 / 10 * 1e3, 3)
            cfg5[pre5 + 'ms_per_forward_one_in_flight'] = round(statistics.median(r51) / 5 * 1e3, 3)
            del p5, e5
            torch.cuda.empty_cache()
        del sd5, imgs5

    # ---- training step (BASELINE config 4's per-GPU batch: 32 images, fp32, the whole network: forward in training mode, 42-term
    #      objective, backward, flat gradient bucket, one AdamW launch -- dir_amd/train/step.py); rank 0, single-GPU runs only
    train = None
    if rank == 0 and world == 1 and half and not args.no_train:
        from dir_amd.optim import FlatAdamW
        from dir_amd.train import step as TSTEP
        torch.cuda.empty_cache()
        TB = 32
        is_buf = lambda k: any(t in k for t in ('running_', 'num_batches', 'mano_layer', 'img_gird', 'seg_loss.weight'))  # noqa: E731
        tparams = {k: torch.nn.Parameter(v.clone().to(dev)) for k, v in sd.items() if not is_buf(k)}
        tbuf = {k: v.clone().to(dev) for k, v in sd.items() if is_buf(k) and 'num_batches' not in k}
        topt = FlatAdamW(list(tparams.values()), lr=1e-5)
        topt.set_inactive(TSTEP.inactive_parameters(tparams))
        rng = np.random.RandomState(0)
        dv = lambda a_: torch.from_numpy(np.ascontiguousarray(a_)).to(dev)  # noqa: E731
        timg = torch.randn(TB, 3, 256, 256, device=dev, generator=g)
        ttar, tmeta = {}, {}
        for s_ in ('left', 'right'):
            ttar['joint_2d_' + s_] = dv(rng.uniform(-1, 1, (TB, 21, 3)).astype(np.float32))
            ttar['mesh_2d_' + s_] = dv(rng.uniform(-1, 1, (TB, 778, 3)).astype(np.float32))
            ttar['joint_3d_' + s_] = dv(rng.normal(0, 0.05, (TB, 21, 3)).astype(np.float32))
            ttar['mesh_3d_' + s_] = dv(rng.normal(0, 0.05, (TB, 778, 3)).astype(np.float32))
            tmeta['center_' + s_] = dv(rng.normal(0, 0.1, (TB, 1, 3)).astype(np.float32))
        ttar['seg'] = dv(rng.randint(0, 3, (TB, 1, 256, 256)).astype(np.float32))
        ttar['dense'] = dv(rng.rand(TB, 3, 256, 256).astype(np.float32))
        tfaces = tuple(dv(synth.loss_faces(s_, 1234).astype(np.int64)) for s_ in ('left', 'right'))
        tt, objective = [], []
        for i in range(4):
            sync()
            t0 = time.perf_counter()
            tl = TSTEP.train_step(tparams, tbuf, timg, ttar, tmeta, tfaces, topt)
            sync()
            tt.append(time.perf_counter() - t0)
            objective.append(round(sum(float(v) for v in tl.values()), 4))
        best = min(tt[1:])
        # executed arithmetic: forward 36.8 GFLOP per image (all convolutions materialised in training) + data and weight gradients = 3x
        train = {'batch_per_gpu': TB, 'seconds_per_step': round(best, 4), 'images_per_sec': round(TB / best, 1), 'steps_timed': 3,
                 'objective_per_step': objective, 'dtype': 'f32 tensors; convolutions (forward, data and weight gradients) in split precision f16x3',
                 'algorithmic_tflops': round(3 * ALG_GFLOP_PER_IMAGE * 1e9 * TB / best / 1e12, 2), 'peak_tflops': PEAK['f16x3'] / 1e12,
                 'frac_of_f16x3_mfma_peak': round(3 * ALG_GFLOP_PER_IMAGE * 1e9 * TB / best / PEAK['f16x3'], 4),
                 'peak_memory_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                 'note': 'dir_amd.train.step.train_step on one GPU (no exchange partner): training-mode forward, 42-term objective, backward, flat '
                         'gradient bucket, one AdamW launch; eager, about 1300 library calls + 300 torch operators per step, every convolution weight packed by one launch (round 2: 0.088 s, round 3: 0.047 s; DESIGN.md section 10)'}
        # the same step with everything but the optimiser replayed as one HIP graph (dir_amd.train.step.GraphedTrainStep: same kernels, same bits):
        # two eager calibration steps, one capture, then timed replays -- how a training loop would run it
        try:
            gstep = TSTEP.GraphedTrainStep(tparams, tbuf, topt, tfaces)
            for _ in range(4):
                gstep(timg, ttar, tmeta)
            sync()
            gt = []
            for _ in range(5):
                t0 = time.perf_counter()
                gstep(timg, ttar, tmeta)
                sync()
                gt.append(time.perf_counter() - t0)
            gmed = statistics.median(gt)
            train['graph_captured_seconds_per_step'] = round(gmed, 4)
            train['graph_captured_images_per_sec'] = round(TB / gmed, 1)
            # the record's headline figures are the graph-captured step's from round 5 on (the eager step is bound by its ~1 500 Python-issued launches
            # as much as by the GPU: 0.030-0.031 s whatever the kernels cost); rounds 2-4 quoted the eager figure, kept beside it
            train['eager_seconds_per_step'], train['eager_images_per_sec'] = train['seconds_per_step'], train['images_per_sec']
            train['seconds_per_step'], train['images_per_sec'] = round(gmed, 4), round(TB / gmed, 1)
            train['algorithmic_tflops'] = round(3 * ALG_GFLOP_PER_IMAGE * 1e9 * TB / gmed / 1e12, 2)
            train['frac_of_f16x3_mfma_peak'] = round(3 * ALG_GFLOP_PER_IMAGE * 1e9 * TB / gmed / PEAK['f16x3'], 4)
            train['how'] = 'GraphedTrainStep: forward, objective, backward and gradient moves replayed as one HIP graph, AdamW after it (same kernels and bits as the eager step; one eager re-calibration step every 50)'
            del gstep
        except Exception as e:          # (the eager figure above stands on its own)
            train['graph_captured_error'] = repr(e)[:200]
        del tparams, tbuf, topt
        torch.cuda.empty_cache()

    # ---- CPU baselines on the host cores (rank 0, single-GPU runs only), bounded samples:
    #   port        the numpy oracle (CPU restatement of the reference).  OpenBLAS is pinned to the thread count that serves these
    #               GEMM sizes best on the box (measured: 8-16 threads 4.2 img/s, 32 threads 2.4, 64 threads 1.1 -- oversubscription)
    #   torch_ops   the same graph with its dense operators (99.9 % of the FLOPs) on stock torch CPU kernels (oracle/torch_ops.py),
    #               torch.set_num_threads(os.cpu_count()), at B = 64 and B = 1 -- what the reference itself does on a CPU
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.dir_forward import dir_forward
        from oracle.torch_ops import stock_torch_dense_ops
        from threadpoolctl import threadpool_limits
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        nthreads = min(args.cpu_threads, ncpu)
        n, chunk = args.cpu_sample, 8
        ximg = img[:min(max(n, 64), B)].cpu().numpy()
        with threadpool_limits(limits=nthreads):
            dir_forward(sd_np, ximg[:1])              # warm-up (BLAS thread pool, page faults)
            t0 = time.perf_counter()
            for i in range(0, n, chunk):
                dir_forward(sd_np, ximg[i % len(ximg):i % len(ximg) + chunk])
            tc = time.perf_counter() - t0
        # torch's CPU kernels at every hardware thread of this box were measured 20x SLOWER than at 16 (256 threads: 0.6 images/s at
        # B = 64, 0.013 at B = 1 -- oversubscribed oneDNN thread pool), so the torch leg runs at the numpy leg's thread count too
        tor = {}
        with stock_torch_dense_ops(nthreads):
            dir_forward(sd_np, ximg[:2])
            for bb, nrep in ((1, 2), (min(64, len(ximg)), 1)):
                t0 = time.perf_counter()
                for _ in range(nrep):
                    dir_forward(sd_np, ximg[:bb])
                tt = time.perf_counter() - t0
                tor['B=%d' % bb] = {'images_per_sec': round(bb * nrep / tt, 3), 'seconds': round(tt, 2), 'forwards': nrep}
        cpu_model = None
        try:
            with open('/proc/cpuinfo') as f:
                cpu_model = next((ln.split(':', 1)[1].strip() for ln in f if ln.startswith('model name')), None)
        except OSError:
            pass
        cpu = {'value': round(n / tc, 3), 'unit': 'images/sec', 'cores': int(nthreads), 'kind': 'port', 'os_cpu_count': int(ncpu),
               'cpu_model': cpu_model,
               'thread_policy': 'min(--cpu-threads=%d, cores this process may run on=%d): OpenBLAS / oneDNN at every hardware thread of the box '
                                'oversubscribe and run 4-20x slower (measured r02: 64 threads 1.1 img/s, 256 threads 0.6) -- pass --cpu-threads N to '
                                'change it' % (args.cpu_threads, ncpu),
               'sample': '%d images in chunks of %d, fp32 forward of oracle/dir_forward.py (numpy + OpenBLAS, %d threads), '
                         '%.1f s' % (n, chunk, nthreads, tc),
               'torch_ops': dict(tor, threads=int(nthreads), note='oracle/dir_forward.py with conv / BN / pool / upsample / linear on stock torch '
                                 'CPU kernels (oracle/torch_ops.py); token path numpy')}

    if rank == 0:
        line = {'metric': 'images/sec at 256x256 bs=%d, 3 stage outputs (DIR.forward eval)' % B, 'value': round(value, 1),
                'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                'dtype': args.dtype, 'data': 'synthetic',
                'config': {'workload': 'BASELINE configs[1]: batch 64 synthetic 256x256 per GPU, ResNet-50 + init '
                                       'regression + 2 refinement stages (3 stage outputs), seg/dense/proj_feat heads; 16-bit storage = %s' % (
                                           'IEEE f16 (the bf16 data path and MFMA rate, 11-bit significands: every stage within 0.01 mm; bf16 in bf16_storage_mode)' if args.dtype == 'f16' else args.dtype),
                           'batch_per_gpu': B, 'graph': not args.no_graph, 'forwards_in_flight': args.inflight,
                           'hip_hw_queues': os.environ.get('GPU_MAX_HW_QUEUES'),
                           'ms_per_forward_one_in_flight': None if serial_ms is None else round(serial_ms, 3),
                           'ms_per_forward_one_in_flight_timed_graphs': None if serial_tp_ms is None else round(serial_tp_ms, 3),
                           'conv_tuning': conv_tuning, 'conv_tuning_check': table_check, 'tunings_bit_identical': tunings_equal, 'weights': 'synthetic (dir_amd.synth seed 1234)',
                           'sharding': 'independent images per GPU, no data-path collective', 'outputs_finite': finite,
                           'overlapped_equals_one_at_a_time': reproducible, 'world_size_observed': world_observed,
                           'backend': ('nccl (RCCL)' if backend == 'nccl' else 'gloo (ranks may share a GPU: code-path test, not a measurement)') if world > 1 else 'none (single process)',
                           'timed_regions': len(regions), 'region_ms_per_step': [round(r / args.steps * 1e3, 3) for r in regions],
                           'region_spread': round((max(regions) - min(regions)) / dt, 4),
                           'statistic': 'median region', 'weights_kind': args.weights,
                           # the line carries its own parity and both 16-bit kinds (the driver keeps `config`)
                           'parity': headline_parity,
                           'images_per_sec_by_mode': {k_: v_ for k_, v_ in (
                               (args.dtype + '_storage (value)', round(value, 1)),
                               (None if other_half is None else other_half['dtype'] + '_storage', None if other_half is None else other_half['images_per_sec']),
                               ('f16x3 (1e-4 mm grade)', None if parity is None else parity['images_per_sec']),
                               ('f32', None if fp32 is None else fp32['images_per_sec'])) if k_ is not None and v_ is not None},
                           'parity_by_mode': {k_: (None if v_ is None or v_.get('parity') is None else {kk: v_['parity'][kk] for kk in ('max_abs_xyz_m', 'mpjpe_mm_max')})
                                              for k_, v_ in (('f16x3', parity), ('f32', fp32), ('other_16bit', other_half)) if v_ is not None}},
                'parity': headline_parity, 'roofline': roof, 'power': power, 'cpu_baseline': cpu, 'fp32_mode': fp32, 'parity_mode_f16x3': parity, 'fp16_mode': f16m, 'fp16_storage_mode' if args.dtype == 'bf16' else 'bf16_storage_mode': other_half, 'pgcn': pgcn, 'train_step': train, 'without_proj_feat': no_pf, 'config5_hrnet': cfg5}
        # the full record (per-kernel tables, notes, sub-mode rooflines) goes to a side file and to stderr; the LAST stdout line is the compact
        # headline (dir_amd/benchline.py: <= 4 KB, every contract key + roofline + cpu_baseline) -- round 3's 20 KB line went unparsed
        from dir_amd import benchline
        detail_path = args.detail_out
        try:
            with open(os.path.join(ROOT, detail_path) if not os.path.isabs(detail_path) else detail_path, 'w') as f:
                json.dump(line, f)
        except OSError as e:
            sys.stderr.write('bench.py: could not write %s: %s\n' % (detail_path, e))
        sys.stderr.write(json.dumps(line) + '\n')
        sys.stderr.flush()
        print(json.dumps(benchline.compact(line, detail_path)), flush=True)
    if world > 1:
        barrier()                                 # rank 0 measured its roofline pass alone: the others leave the group together with it
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
