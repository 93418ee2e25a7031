"""Build-owned synthetic parameters for the DIR hot path.

Neither the licensed MANO pkl files nor the published checkpoint are available
(SURVEY.md section 0 / 8c), so every parity test, golden vector and benchmark
runs on parameters regenerated *by name* from a counter-based RNG (numpy
Philox keyed by (seed, crc32(name))).  The golden generator
(oracle/gen_golden.py, which imports the reference) and the GPU-side tests call
the same functions, so only inputs/outputs are committed as fixtures - never
weights.

Distributions follow the reference's own initialisers where it has one
(models/dir.py:77-84,248-257: conv N(0, sqrt(2/(k*k*Cout))), Linear N(0,1e-3);
SemGCN/p_graph_conv.py:19-35: xavier_uniform gain 1.414), perturbed so that
no operator degenerates to a no-op (BatchNorm statistics, e_1 edge logits and
spatial_pos_embed are constant in the reference's init).
"""
import zlib

import numpy as np

MANO_PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]


def rng_for(name, seed=1234):
    return np.random.Generator(np.random.Philox(key=[int(seed), zlib.crc32(name.encode())]))


def synthetic_mano_tables(side, seed=1234):
    """Synthetic MANO tables with the real shapes/dtypes (cf. the buffers registered at
    manopth/manopth/manolayer.py:71-98 and the kintree at models/manolayer.py:67)."""
    g = rng_for('mano_tables.' + side, seed)
    f64 = np.float64
    v_template = (g.uniform(-1, 1, (778, 3)) * np.array([0.09, 0.05, 0.02])).astype(f64)
    shapedirs = g.normal(0, 0.004, (778, 3, 10)).astype(f64)
    posedirs = g.normal(0, 0.0015, (778, 3, 135)).astype(f64)
    # skinning weights: each vertex influenced by up to 4 joints, rows sum to 1
    weights = np.zeros((778, 16), f64)
    for v in range(778):
        idx = g.choice(16, size=4, replace=False)
        w = g.uniform(0.05, 1.0, 4)
        weights[v, idx] = w / w.sum()
    # joint regressor: each joint a convex combination of 24 vertices
    J_regressor = np.zeros((16, 778), f64)
    for j in range(16):
        idx = g.choice(778, size=24, replace=False)
        w = g.uniform(0.1, 1.0, 24)
        J_regressor[j, idx] = w / w.sum()
    q, _ = np.linalg.qr(g.normal(0, 1, (45, 45)))
    hands_components = (q * g.uniform(0.3, 1.2, (45, 1))).astype(f64)
    hands_mean = g.normal(0, 0.15, (45,)).astype(f64)
    betas = np.zeros((10,), f64)
    faces = g.integers(0, 778, (1538, 3)).astype(np.uint32)
    kintree = np.array([[4294967295 if p < 0 else p for p in MANO_PARENTS], list(range(16))], dtype=np.int64)
    return dict(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs, weights=weights,
                J_regressor=J_regressor, hands_components=hands_components, hands_mean=hands_mean,
                betas=betas, f=faces, kintree_table=kintree)


def mano_buffers(side, seed=1234, ncomps=45, flat_hand_mean=False):
    """The th_* buffers a ManoLayer registers (manopth/manopth/manolayer.py:71-98), float32."""
    t = synthetic_mano_tables(side, seed)
    f32 = np.float32
    hm = np.zeros(45) if flat_hand_mean else t['hands_mean']
    return {
        'th_betas': t['betas'].astype(f32)[None],
        'th_shapedirs': t['shapedirs'].astype(f32),
        'th_posedirs': t['posedirs'].astype(f32),
        'th_v_template': t['v_template'].astype(f32)[None],
        'th_J_regressor': t['J_regressor'].astype(f32),
        'th_weights': t['weights'].astype(f32),
        'th_faces': t['f'].astype(np.int32).astype(np.int64),
        'th_hands_mean': hm.astype(f32)[None],
        'th_comps': t['hands_components'].astype(f32),
        'th_selected_comps': t['hands_components'][:ncomps].astype(f32),
    }


_REGRESSOR_LEAVES = ('mano_left', 'mano_right', 'offset')


def _is_regressor(name):
    parts = name.split('.')
    return len(parts) >= 2 and parts[-2] in _REGRESSOR_LEAVES and \
        ('init_regressor' in parts or 'regressor' in parts)


def _branch_final(name):
    """conv weights whose output is ADDED to a residual stream (or is a plain projection without ReLU): Bottleneck conv3 / downsample,
    hourglass Residual conv3 / skip_layer, and the second conv of the Sequential heads"""
    return (name.endswith('.conv3.weight') or name.endswith('.conv3.conv.weight') or name.endswith('.skip_layer.conv.weight')
            or name.endswith('.downsample.0.weight') or name.endswith('.3.weight'))


def synth_tensor(name, shape, shapes, seed=1234, cond=False):
    """Value for state-dict entry `name` (shape `shape`); `shapes` maps every key to its shape.
    cond = True: the "trained-like" flavour.  The default flavour follows the reference's initialisers, whose
    fan-OUT convolution scale amplifies every channel-reducing 1x1 layer (x8 per bottleneck entry) while the random BatchNorm statistics
    do not renormalise: |c4| ~ 2e2, decoder activations ~ 1e3, training-mode seg logits ~ 1e5 -- nothing a trained network shows, and
    bad conditioning for parity gates (bf16 init-stage error 0.12 mm, fp32 gradients good to 1e-2 only).  Here convolutions are scaled by
    fan-IN (variance preserving through ReLU), branch-final convolutions by half of that, so activations stay O(1) through all 53 + 20
    layers as in a trained, BatchNorm-normalised network; the regressors use the reference's own N(0, 1e-3)."""
    g = rng_for(name, seed)
    shape = tuple(shape)
    base, _, leaf = name.rpartition('.')
    f32 = np.float32
    if leaf == 'num_batches_tracked':
        return np.zeros(shape, np.int64)
    if leaf == 'img_gird':
        S = int(round(np.sqrt(shape[0])))
        c = np.arange(S, dtype=f32) + 0.5
        gx, gy = np.meshgrid(c, c, indexing='ij')
        return np.stack((gy, gx), -1).reshape(S * S, 2).astype(f32)
    if name == 'seg_loss.weight':
        return np.array([.1, .45, .45], f32)
    if leaf.startswith('th_'):
        side = 'left' if 'mano_layer_left' in base else 'right'
        return mano_buffers(side, seed)[leaf]
    if leaf == 'running_mean':
        return g.normal(0, 0.1, shape).astype(f32)
    if leaf == 'running_var':
        return g.uniform(0.5, 1.5, shape).astype(f32)
    is_bn = (base + '.running_mean') in shapes
    last = base.split('.')[-1]
    is_ln = (not is_bn) and len(shape) == 1 and (last.startswith('norm') or last == 'spatial_norm'
                                                 or base.endswith('head.0'))
    if is_bn or is_ln:
        if leaf == 'weight':
            if cond and (('.branches.' in name and '.bn2.' in name) or '.fuse_layers.' in name):
                # HRNet-W48 (f4): 32 BasicBlocks in sequence per branch and 8 fuse sums -- residual branches and fused terms enter with a small
                # gain (what zero-init-residual training leaves behind), or the trained-like flavour would not stay O(1)
                return g.uniform(0.15, 0.3, shape).astype(f32)
            return g.uniform(0.8, 1.2, shape).astype(f32)
        return g.normal(0, 0.1, shape).astype(f32)
    if leaf == 'W':  # PGraphConv per-node weights [2,J,Cin,Cout], xavier_uniform gain 1.414
        fan_in, fan_out = shape[1] * shape[2] * shape[3], shape[0] * shape[2] * shape[3]
        a = 1.414 * np.sqrt(6.0 / (fan_in + fan_out))
        # the reference's xavier bound for this 4-D shape is tiny (fan over J*C*C); enlarge so the
        # GCN output is not numerically negligible next to its bias
        a = max(a, 1.0 / np.sqrt(shape[2]))
        return g.uniform(-a, a, shape).astype(f32)
    if leaf in ('e_0', 'e_1'):
        return g.normal(1.0, 0.5, shape).astype(f32)
    if leaf == 'spatial_pos_embed':
        return g.normal(0, 0.02, shape).astype(f32)
    if leaf == 'weight':
        if len(shape) == 4:
            co, ci, kh, kw = shape
            if cond:
                fan_in = kh * kw * ci
                return g.normal(0, np.sqrt((0.5 if _branch_final(name) else 2.0) / fan_in), shape).astype(f32)
            return g.normal(0, np.sqrt(2.0 / (kh * kw * co)), shape).astype(f32)
        if len(shape) == 3:
            co, ci, k = shape
            b = 1.0 / np.sqrt(ci * k)
            return g.uniform(-b, b, shape).astype(f32)
        if len(shape) == 2:
            if _is_regressor(name):
                # c4 of a random-init ResNet-50 has |x|~2e2, the refinement tokens |x|~1: keep the Linear
                # contribution below the bias so joints project inside the image and the next stage's
                # gather / rasteriser see real work
                sig = 1e-3 if cond else 2e-5 if 'init_regressor' in name else 2e-4      # cond: models/dir.py:256-257,336-337
                return g.normal(0, sig, shape).astype(f32)
            b = 1.0 / np.sqrt(shape[1])
            return g.uniform(-b, b, shape).astype(f32)
    if leaf == 'bias':
        if _is_regressor(name):
            n = shape[0]
            if n == 64:   # [0:6] 6D root rot | [6:51] pose PCA | [51:61] betas | [61] scale | [62:64] trans
                v = np.zeros(64, f32)
                v[0:6] = np.array([1, 0, 0, 0, 1, 0], f32) + g.normal(0, 0.3, 6)
                v[6:51] = g.normal(0, 0.4, 45)
                v[51:61] = g.normal(0, 0.5, 10)
                v[61] = 5.0 + g.normal(0, 0.5)
                v[62:64] = g.normal(0, 0.15, 2)
                return v.astype(f32)
            return g.normal(0, 0.3, shape).astype(f32)
        w = shapes.get(base + '.weight')
        fan_in = int(np.prod(w[1:])) if w is not None and len(w) > 1 else shape[0]
        b = 1.0 / np.sqrt(fan_in)
        return g.uniform(-b, b, shape).astype(f32)
    raise KeyError('no synthetic rule for state-dict entry %r shape %r' % (name, shape))


def synth_state_dict(shapes, seed=1234, cond=False):
    """shapes: {key: shape}.  Returns {key: np.ndarray}.  cond: the trained-like flavour (synth_tensor)."""
    return {k: synth_tensor(k, s, shapes, seed, cond) for k, s in shapes.items()}


def synth_input(name, shape, seed=1234, kind='normal', lo=-1.0, hi=1.0, scale=1.0):
    g = rng_for('input.' + name, seed)
    if kind == 'normal':
        return (g.normal(0, 1, shape) * scale).astype(np.float32)
    return g.uniform(lo, hi, shape).astype(np.float32)


def loss_faces(side, seed=1234):
    """Triangle table for the mesh losses (models/loss.py): the synthetic `f` with its few degenerate rows (a random index drawn
    twice) repaired -- on a zero-area triangle the reference's normal is normalised rounding noise, so nothing can be pinned."""
    f = np.array(synthetic_mano_tables(side, seed)['f'], np.int64)
    for i in range(f.shape[0]):
        a, b, c = f[i]
        if a == b or a == c or b == c:
            f[i] = [a, (a + 1 + i % 7) % 778, (a + 9 + i % 11) % 778]
    assert not ((f[:, 0] == f[:, 1]) | (f[:, 0] == f[:, 2]) | (f[:, 1] == f[:, 2])).any()
    return f
