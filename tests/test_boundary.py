"""CPU-only: the drop-in boundary.  The dir_amd modules expose the reference's class names, constructor
signatures and -- key for loading a reference checkpoint -- exactly the reference's state-dict keys and shapes."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def manifest(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def test_dir_state_dict_matches_reference_manifest():
    from dir_amd.models.dir import DIR
    net = DIR(21, './misc/mano', 0)
    ours = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    ref = manifest('manifest_dir.json')
    assert len(ref) == 963
    assert sorted(ours) == sorted(ref)
    assert ours == ref
    # a state dict with the reference's keys loads strictly (what apps/eval.py:107-108 does with strict=False)
    from dir_amd import synth
    vals = synth.synth_state_dict(ref, 1234)
    net.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in vals.items()}, strict=True)
    assert hasattr(net.init_regressor.mano_layer_left, 'th_faces')        # read by train.py / models/dir.py:507-510


def test_stage_state_dict_matches_reference_manifest():
    from dir_amd.models.dir import Joint2BoneFeature
    for S, dist in ((16, 1), (32, 2)):
        m = Joint2BoneFeature(256, 128, 64, 21, S, 'unused', 0, distance=dist)
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == manifest('manifest_stage%d.json' % S)
        c = np.arange(S, dtype=np.float32) + 0.5
        assert np.allclose(m.img_gird.numpy()[:3], [[c[0], c[0]], [c[1], c[0]], [c[2], c[0]]])   # (x+0.5, y+0.5)


def test_reference_call_signatures():
    import inspect
    from dir_amd.manopth.manolayer import ManoLayer
    from dir_amd.models.dir import DIR
    from dir_amd.SemGCN.p_gcn import ResSimplePGCN
    from dir_amd.SemGCN.p_graph_conv import PGraphConv
    from dir_amd.SemGCN.utils import adj_mx_from_edges, get_sketch_setting
    from dir_amd.transformer.mixSTE import STE
    assert list(inspect.signature(DIR.__init__).parameters)[:4] == ['self', 'joint_num', 'mano_path', 'root_joint']
    assert list(inspect.signature(DIR.forward).parameters) == ['self', 'input', 'target', 'meta_info']
    assert list(inspect.signature(ManoLayer.forward).parameters) == ['self', 'th_pose_coeffs', 'th_betas', 'th_trans',
                                                                     'root_palm', 'share_betas']
    assert list(inspect.signature(PGraphConv.__init__).parameters) == ['self', 'in_features', 'out_features', 'adj', 'bias']
    assert list(inspect.signature(STE.__init__).parameters)[:5] == ['self', 'num_joints', 'in_chans', 'out_dim', 'depth']
    adj = adj_mx_from_edges(21, get_sketch_setting(), sparse=False, eye=False)
    assert adj.shape == (21, 21) and int((adj > 0).sum()) == 40 and float(adj[0].sum()) == pytest.approx(1.0)
    g = ResSimplePGCN(adj, 128, num_layers=4)
    assert g.gconv_layers[0].gconv.W.shape == (2, 21, 128, 128) and g.gconv_layers[0].gconv.e_1.shape == (1, 40)
    with pytest.raises(NotImplementedError):
        PGraphConv(64, 64, adj)


def test_next_row_call_signatures():
    """8f rank 1 mirrors: models/manolayer.py::ManoLayer (:100-105, :251) and apps/eval.py::Jr (:22-44)"""
    import inspect
    from dir_amd.apps.eval import Jr
    from dir_amd.models.manolayer import ManoLayer as GTManoLayer
    assert list(inspect.signature(GTManoLayer.__init__).parameters)[:5] == ['self', 'manoPath', 'center_idx', 'use_pca', 'new_skel']
    assert list(inspect.signature(GTManoLayer.forward).parameters) == ['self', 'root_rotation', 'pose', 'shape', 'trans', 'scale']
    assert list(inspect.signature(Jr.__init__).parameters) == ['self', 'J_regressor', 'device']
    layer = GTManoLayer.synthetic('left', center_idx=None)
    assert layer.J_regressor.shape == (16, 778) and layer.hands_components_inv.shape == (45, 45)
    assert layer.parent == [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14] and layer.get_faces().shape == (1538, 3)
    assert list(layer.state_dict().keys()) == ['hands_components', 'hands_components_inv']      # the persistent buffers


def test_gpu_only_in_both_modes():
    """no CPU fallback: eval and training mode alike refuse to run without the GPU (the reference calls .cuda() itself, models/dir.py:514)"""
    from dir_amd.models.dir import DIR
    net = DIR(21, 'unused', 0)
    for mode in (net.train, net.eval):
        with pytest.raises((RuntimeError, AssertionError, Exception)) as e:
            mode()({'img': torch.zeros(1, 3, 256, 256)}, None, None)
        assert not isinstance(e.value, NotImplementedError)


def test_csr_adjacency_constants_match_edge_order(golden):
    """the __constant__ CSR tables in dir_amd/csrc/tokens.hip index e_1 in the reference's row-major nonzero order."""
    import re
    src = open(os.path.join(os.path.dirname(GOLDEN), '..', 'dir_amd', 'csrc', 'tokens.hip')).read()
    off = [int(v) for v in re.search(r'kNbrOff\[22\] = \{([^}]*)\}', src).group(1).split(',')]
    idx = [int(v) for v in re.search(r'kNbrIdx\[40\] = \{([^}]*)\}', src).group(1).split(',')]
    order = golden('g2_pgcn')['edge_order']
    rows = [r for r in range(21) for _ in range(off[r + 1] - off[r])]
    assert np.array_equal(np.stack([rows, idx], 1), order)


def test_reference_import_lines_resolve_through_the_compat_shim():
    """apps/eval.py:15-19 and models/dir.py:7-15 import `models.dir`, `models.manolayer`, `SemGCN.*`, `transformer.mixSTE`,
    `manopth.manolayer` as top-level packages: with dir_amd/compat on sys.path those lines work unchanged and give the mirrors."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = '''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from models.dir import DIR
from models.manolayer import ManoLayer
from models.backbone.hourglass import Residual
from models.backbone.resnet import resnet50 as ResNet50
from SemGCN.utils import adj_mx_from_edges, get_sketch_setting
from SemGCN.p_gcn import ResSimplePGCN
from SemGCN.p_graph_conv import PGraphConv
from transformer.mixSTE import STE
from manopth.manolayer import ManoLayer as ObmanManoLayer
import dir_amd.models.dir, dir_amd.manopth.manolayer, dir_amd.models.manolayer
assert DIR is dir_amd.models.dir.DIR and ObmanManoLayer is dir_amd.manopth.manolayer.ManoLayer
assert ManoLayer is dir_amd.models.manolayer.ManoLayer
assert ResNet50().inplanes == 2048
print("ok")
''' % (root, os.path.join(root, 'dir_amd', 'compat'))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == 'ok', r.stderr
