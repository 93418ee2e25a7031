"""Drop-in for the reference's models/manolayer.py::ManoLayer (the ground-truth MANO layer): same
constructor signature, buffers (hands_components, hands_components_inv, J_regressor, J_zero, weights, posedirs, v_template,
shapedirs, hands_mean), `faces`, `parent`, `new_order`, and forward(root_rotation, pose, shape, trans=None, scale=None)
-> (verts [B,778,3], joints [B,21,3]).  forward runs dir_gt_mano_forward (one launch) instead of ~150 ATen calls and a
serial Python chain; dataset/interhand.py:130-149 calls it per sample on the CPU, here whole batches go to the GPU.

`manoPath` is read like the reference does (pickle with numpy / scipy members, i.e. the output of the reference's
convert_mano_pkl); when it is None or the licensed file is absent use ManoLayer.synthetic(side).  Only forward() is built:
the inverse helpers (Rmat2axis, axis2pca, get_local_frame) are dataset-authoring utilities outside the evaluated path.
"""
import os
import pickle

import numpy as np
import torch
from torch.nn import Module

from .. import _capi, synth
from ..engine import _pad_rows


class ManoLayer(Module):
    def __init__(self, manoPath, center_idx=9, use_pca=True, new_skel=False, _tables=None):
        super().__init__()
        self.center_idx, self.use_pca, self.new_skel = center_idx, use_pca, new_skel
        if _tables is not None:
            manoData = _tables
        elif manoPath is not None and os.path.isfile(manoPath):
            with open(manoPath, 'rb') as f:
                manoData = pickle.load(f, encoding='latin1')                     # models/manolayer.py:106
        else:
            raise FileNotFoundError('%r: the licensed MANO pickle is not distributed; use ManoLayer.synthetic(side) for '
                                    'synthetic tables' % (manoPath,))
        self.new_order = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32)))  # noqa: E731
        comps = f32(manoData['hands_components'])
        self.register_buffer('hands_components', comps)
        self.register_buffer('hands_components_inv', torch.inverse(comps))
        J = manoData['J_regressor']
        J = J.toarray() if hasattr(J, 'toarray') else np.asarray(J)
        self.register_buffer('J_regressor', f32(J), persistent=False)
        self.register_buffer('J_zero', f32(manoData['J'] if 'J' in manoData else J @ np.asarray(manoData['v_template'])),
                             persistent=False)
        self.register_buffer('weights', f32(manoData['weights']), persistent=False)
        self.register_buffer('posedirs', f32(manoData['posedirs']), persistent=False)
        self.register_buffer('v_template', f32(manoData['v_template']), persistent=False)
        sdirs = manoData['shapedirs']
        self.register_buffer('shapedirs', f32(sdirs if isinstance(sdirs, np.ndarray) else sdirs.r.copy()), persistent=False)
        self.register_buffer('hands_mean', f32(manoData['hands_mean']), persistent=False)
        self.faces = manoData['f']
        self.parent = [-1] + [int(manoData['kintree_table'][0, i]) for i in range(1, 16)]
        self._packed = None

    @classmethod
    def synthetic(cls, side, center_idx=9, use_pca=True, new_skel=False, seed=1234):
        """the layer over dir_amd.synth's synthetic MANO tables (shapes / dtypes of the real asset)"""
        t = synth.synthetic_mano_tables(side, seed)
        t = dict(t, J=t['J_regressor'] @ t['v_template'])
        return cls(None, center_idx, use_pca, new_skel, _tables=t)

    def get_faces(self):
        return self.faces

    def _tables(self, dev):
        key = (dev, tuple((b.data_ptr(), b._version) for b in (self.shapedirs, self.posedirs, self.v_template, self.J_regressor,
                                                                self.weights, self.hands_mean, self.hands_components)))
        if self._packed is None or self._packed[0] != key:
            f = lambda b: b.detach().to(dev).float()  # noqa: E731
            t = dict(shapedirs_t=_pad_rows(f(self.shapedirs).reshape(2334, 10).t()),
                     posedirs_t=_pad_rows(f(self.posedirs).reshape(2334, 135).t()),
                     v_template=f(self.v_template).reshape(2334).contiguous(),
                     j_template=(f(self.J_regressor).double() @ f(self.v_template).double()).float().contiguous(),
                     j_shapedirs=torch.einsum('jv,vck->jck', f(self.J_regressor).double(), f(self.shapedirs).double()).float().contiguous(),
                     weights=f(self.weights).contiguous(), hands_mean=f(self.hands_mean).reshape(45).contiguous(),
                     comps=f(self.hands_components).contiguous())
            T = _capi.ManoTables(t['shapedirs_t'].data_ptr(), t['posedirs_t'].data_ptr(), t['v_template'].data_ptr(),
                                 t['j_template'].data_ptr(), t['j_shapedirs'].data_ptr(), t['weights'].data_ptr(),
                                 t['hands_mean'].data_ptr(), t['comps'].data_ptr(), 0, -1, 0)
            self._packed = (key, T, t)
        return self._packed[1]

    def forward(self, root_rotation, pose, shape, trans=None, scale=None):
        _capi.require_cuda(root_rotation, pose, shape, trans, scale)
        bs = root_rotation.shape[0]
        R = _capi.f32c(root_rotation.reshape(bs, 9))
        if self.use_pca:
            ncomps = pose.shape[1]
            P = _capi.f32c(pose)
        else:
            ncomps = 0
            P = _capi.f32c(pose.reshape(bs, 135))
        S = _capi.f32c(shape)
        T = None if trans is None else _capi.f32c(trans.reshape(bs, 3))
        C_ = None if scale is None else _capi.f32c(scale.reshape(bs))
        verts = torch.empty(bs, 778, 3, device=R.device)
        joints = torch.empty(bs, 21, 3, device=R.device)
        with torch.cuda.device(R.device):
            _capi.check(_capi.lib().dir_gt_mano_forward(self._tables(R.device), _capi.ptr(R), _capi.ptr(P), ncomps, _capi.ptr(S),
                                                        _capi.ptr(T), _capi.ptr(C_), -1 if self.center_idx is None else int(self.center_idx),
                                                        1 if self.new_skel else 0, _capi.ptr(verts), _capi.ptr(joints), bs,
                                                        _capi.stream_ptr()), 'dir_gt_mano_forward')
        return verts, joints
