"""Foreign kernels beside a running forward (the 64x128 three-buffer-ring convolution tile once corrupted packed-FP32 instructions
of OTHER kernels resident on the same CU; that tile is no longer selected unless DIR_RING_64x128=1).  A torch fused-multiply-add kernel --
compiled by someone else, free to use v_pk_fma_f32 -- runs on a second stream while forwards replay on the slot's stream: both sides
must reproduce their stand-alone results bit for bit, every round."""
import json
import os

import numpy as np
import pytest
import torch

from dir_amd import synth
from dir_amd.engine import DirEngine, ForwardPipeline

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize('tuning,dt', [('heuristic', torch.bfloat16), ('throughput', torch.bfloat16), ('throughput', torch.float16)])
def test_foreign_fma_kernel_beside_the_forward_is_bit_exact_on_both_sides(tuning, dt):
    with open(os.path.join(HERE, 'golden', 'manifest_dir.json')) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
    eng = DirEngine(sd, dtype=dt)                           # torch.float16: f16 storage, bench.py's headline mode
    gen = torch.Generator(device='cuda').manual_seed(3)
    img = torch.randn(64, 3, 256, 256, device='cuda', generator=gen)
    if tuning == 'throughput':                             # the shipped table's kernel mix (256 x 256 tiles, halo reuse, streaming 1x1) beside the foreign kernel
        eng.forward(img)
        eng.autotune(img, reps=1)
        assert eng.load_tuning_table(img, 'gfx950_bf16_b64_throughput') is not None
    pipe = ForwardPipeline(eng, [img])
    pipe.launch(0)
    o = pipe.wait(0)
    keys = ('pd_mesh_xyz_left', 'pd_joint_uv_right', 'pd_offset')
    ref = [o[s][k].clone() for s in range(3) for k in keys] + [o[3]['seg'].clone()]
    a, b, c = (torch.randn(1 << 24, device='cuda', generator=gen) for _ in range(3))
    want = torch.addcmul(c, a, b)                       # a * b + c: one fused multiply-add per element
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for rnd in range(8):
        pipe.launch(0)
        outs = []
        with torch.cuda.stream(side):
            for _ in range(40):                          # ~ the duration of the forward
                outs.append(torch.addcmul(c, a, b))
        o = pipe.wait(0)
        side.synchronize()
        got = [o[s][k] for s in range(3) for k in keys] + [o[3]['seg']]
        assert all(torch.equal(x, y) for x, y in zip(got, ref)), 'forward changed beside a foreign kernel (round %d)' % rnd
        assert all(torch.equal(x, want) for x in outs), 'foreign kernel corrupted beside the forward (round %d)' % rnd


@pytest.mark.parametrize('tuning,dt', [('time', torch.bfloat16), ('throughput', torch.bfloat16), ('throughput', torch.float16)])
def test_four_forwards_in_flight_reproduce_the_one_at_a_time_results(tuning, dt):
    """bench.py's default concurrency (four pipeline slots), with the live time-tuned kernel choice and with the shipped throughput table
    (large tiles on few CUs beside other forwards' kernels: another mix of co-resident kernels): every slot, every round, equals its stand-alone
    result bit for bit, and the two tables agree with each other"""
    with open(os.path.join(HERE, 'golden', 'manifest_dir.json')) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.synth_state_dict(shapes, 1234).items()}
    eng = DirEngine(sd, dtype=dt)                           # torch.float16: f16 storage, bench.py's headline mode
    gen = torch.Generator(device='cuda').manual_seed(11)
    imgs = [torch.randn(64, 3, 256, 256, device='cuda', generator=gen) for _ in range(4)]
    eng.autotune(imgs[0])
    keys = ('pd_mesh_xyz_left', 'pd_joint_uv_right', 'pd_offset')
    o = eng.forward(imgs[0])
    eager = [o[i][k].clone() for i in range(3) for k in keys] + [o[3]['seg'].clone()]
    if tuning == 'throughput':
        assert eng.load_tuning_table(imgs[0], 'gfx950_bf16_b64_throughput') is not None
    pipe = ForwardPipeline(eng, imgs)
    ref = []
    for s in range(4):                                   # one at a time
        pipe.launch(s)
        o = pipe.wait(s)
        ref.append([o[i][k].clone() for i in range(3) for k in keys] + [o[3]['seg'].clone(), o[3]['proj_feat'].clone()])
    assert all(torch.equal(a, b) for a, b in zip(eager, ref[0][:-1]))          # (slot 0 holds imgs[0]: the time-tuned eager forward's bits)
    for rnd in range(5):
        for s in range(4):
            pipe.launch(s)
        for s in range(4):
            o = pipe.wait(s)
            got = [o[i][k] for i in range(3) for k in keys] + [o[3]['seg'], o[3]['proj_feat']]
            assert all(torch.equal(a, b) for a, b in zip(got, ref[s])), (rnd, s)
