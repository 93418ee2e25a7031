"""Drop-in for the reference's `manopth.manolayer.ManoLayer` (manopth/manopth/manolayer.py:13-270):
same constructor keywords, same public `th_*` buffers, same `forward` signature and return value,
computed by ONE fused HIP kernel (dir_mano_forward, dir_amd/csrc/mano.hip) instead of ~4000 ATen ops.

Supported configuration = the one the network uses (models/dir.py:221-224,315-318): root_rot_mode='6D',
joint_rot_mode='axisang', use_pca=True, robust_rot=True.  Other modes raise NotImplementedError: they
are not on the DIR hot path.  When an input requires grad the outputs carry ONE autograd node whose backward is
dir_mano_backward_pair (round 5); otherwise no graph is built.

MANO tables: the licensed MANO_{LEFT,RIGHT}.pkl needs chumpy to unpickle and is out of scope
(SURVEY.md 2); the published checkpoint already carries every th_* buffer, so
`load_state_dict` is the supported way to get real tables.  Without a checkpoint the buffers are filled
with the synthetic tables of dir_amd.synth (flag `synthetic_tables`).
"""
import os

import torch
from torch.nn import Module

from .. import _capi, synth
from ..engine import _pad_rows


class ManoLayer(Module):
    def __init__(self, center_idx=None, flat_hand_mean=True, ncomps=6, side='right', mano_root='mano/models',
                 use_pca=True, root_rot_mode='axisang', joint_rot_mode='axisang', robust_rot=False,
                 check_reflection=True, seed=1234):
        super().__init__()
        if not (root_rot_mode == '6D' and use_pca and robust_rot):
            raise NotImplementedError('dir_amd ManoLayer implements the configuration DIR uses '
                                      "(root_rot_mode='6D', use_pca=True, robust_rot=True); got %r/%r/%r"
                                      % (root_rot_mode, use_pca, robust_rot))
        if ncomps != 45:
            raise NotImplementedError('ncomps must be 45 (models/dir.py:222)')
        if side not in ('right', 'left'):
            raise ValueError('side must be right or left')
        self.center_idx, self.robust_rot, self.rot = center_idx, robust_rot, 6
        self.flat_hand_mean, self.side, self.use_pca = flat_hand_mean, side, use_pca
        self.joint_rot_mode, self.root_rot_mode, self.ncomps = joint_rot_mode, root_rot_mode, ncomps
        self.mano_path = os.path.join(mano_root, 'MANO_RIGHT.pkl' if side == 'right' else 'MANO_LEFT.pkl')
        self.check_reflection = check_reflection
        self.synthetic_tables = True
        for k, v in synth.mano_buffers(side, seed, ncomps, flat_hand_mean).items():
            self.register_buffer(k, torch.from_numpy(v))
        self.kintree_parents = [4294967295, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]
        self._packed = None
        self._packed_key = None

    # -- tables in the kernel's k-major layout, re-packed whenever a th_* buffer changes
    def _tables(self):
        bufs = (self.th_shapedirs, self.th_posedirs, self.th_v_template, self.th_J_regressor, self.th_weights,
                self.th_hands_mean, self.th_selected_comps)
        key = tuple((b.data_ptr(), b._version, str(b.device)) for b in bufs)
        if key != self._packed_key:
            _capi.require_cuda(*bufs)
            f = _capi.f32c
            self._packed = dict(
                shapedirs_t=_pad_rows(self.th_shapedirs.float().reshape(2334, 10).t()),
                posedirs_t=_pad_rows(self.th_posedirs.float().reshape(2334, 135).t()),
                v_template=f(self.th_v_template.reshape(2334)),
                j_template=(self.th_J_regressor.double() @ self.th_v_template.double().reshape(778, 3)).float().contiguous(),
                j_shapedirs=torch.einsum('jv,vck->jck', self.th_J_regressor.double(),
                                         self.th_shapedirs.double()).float().contiguous(),
                weights=f(self.th_weights),
                hands_mean=f(self.th_hands_mean.reshape(45)), comps=f(self.th_selected_comps))
            self._packed_key = key
        return self._packed

    def c_tables(self, center_idx, root_palm=False):
        p = self._tables()
        return _capi.ManoTables(p['shapedirs_t'].data_ptr(), p['posedirs_t'].data_ptr(), p['v_template'].data_ptr(),
                                p['j_template'].data_ptr(), p['j_shapedirs'].data_ptr(), p['weights'].data_ptr(), p['hands_mean'].data_ptr(),
                                p['comps'].data_ptr(), 0 if self.side == 'right' else 1,
                                -1 if center_idx is None else int(center_idx), int(bool(root_palm)))

    def _grad_forward(self, th_pose_coeffs, th_betas, th_trans, root_palm, share_betas):
        """the same forward with an autograd node behind it (the reference's layer is differentiable,
        manopth/manopth/manolayer.py:110-270 under train.py:64-70): dir_mano_forward, and dir_mano_backward_pair for the gradients w.r.t.
        th_pose_coeffs [B,51] and th_betas [B,10] (the forward is recomputed inside the backward kernel, nothing is saved but the inputs)"""
        from .. import functional as F
        if share_betas:
            raise NotImplementedError('share_betas is not differentiated here (not on the DIR path: models/dir.py never passes it)')
        B = th_pose_coeffs.shape[0]
        own_betas = th_betas is None or th_betas.numel() == 1
        betas_in = self.th_betas.expand(B, 10).contiguous() if own_betas else th_betas.to(th_pose_coeffs.device)
        use_trans = th_trans is not None and bool(torch.norm(th_trans) != 0)
        layer = self

        class _Mano(torch.autograd.Function):
            @staticmethod
            def forward(ctx, pose, betas):
                verts, joints = layer._raw(pose.detach(), betas.detach(), use_trans, root_palm)
                ctx.save_for_backward(pose.detach(), betas.detach())
                ctx.set_materialize_grads(False)
                return verts, joints

            @staticmethod
            def backward(ctx, g_verts, g_joints):
                pose, betas = ctx.saved_tensors
                para = torch.cat([_capi.f32c(pose), _capi.f32c(betas), torch.zeros(B, 3, device=pose.device)], 1).contiguous()     # the 64-vector layout of the kernel
                t = layer.c_tables(None if use_trans else layer.center_idx, root_palm)
                with torch.cuda.device(pose.device):
                    g = F.mano_backward([t], [para], g_verts=None if g_verts is None else [g_verts.contiguous()],
                                        g_joints=None if g_joints is None else [g_joints.contiguous()])[0]
                return g[:, :51].contiguous(), (g[:, 51:61].contiguous() if ctx.needs_input_grad[1] else None)
        verts, joints = _Mano.apply(th_pose_coeffs, betas_in)
        if use_trans:                         # (translation: plain torch adds on the node's outputs, differentiated by autograd itself)
            joints = joints + th_trans.unsqueeze(1)
            verts = verts + th_trans.unsqueeze(1)
        return verts, joints

    def forward(self, th_pose_coeffs, th_betas=torch.zeros(1), th_trans=None, root_palm=False, share_betas=False):
        _capi.require_cuda(th_pose_coeffs)
        if torch.is_grad_enabled() and (th_pose_coeffs.requires_grad or (torch.is_tensor(th_betas) and th_betas.requires_grad)):
            return self._grad_forward(th_pose_coeffs, th_betas, th_trans, root_palm, share_betas)
        B = th_pose_coeffs.shape[0]
        pose = th_pose_coeffs.detach()
        if pose.dtype != torch.float32 or pose.stride(-1) != 1:
            pose = _capi.f32c(pose)
        if th_betas is None or th_betas.numel() == 1:
            betas = self.th_betas.expand(B, 10).contiguous()
        else:
            betas = th_betas.detach().to(pose.device)
            if share_betas:
                betas = betas.mean(0, keepdim=True).expand(B, 10)
            if betas.dtype != torch.float32 or betas.stride(-1) != 1 or (B > 1 and betas.stride(0) == 0):
                betas = _capi.f32c(betas)
        use_trans = th_trans is not None and bool(torch.norm(th_trans) != 0)
        verts, joints = self._raw(pose, betas, use_trans, root_palm)
        if use_trans:
            joints = joints + th_trans.unsqueeze(1)
            verts = verts + th_trans.unsqueeze(1)
        return verts, joints

    def _raw(self, pose, betas, use_trans, root_palm):
        """the kernel call: pose [B,51] / betas [B,10] fp32 (detached) -> verts, joints before the optional translation"""
        B = pose.shape[0]
        if pose.dtype != torch.float32 or pose.stride(-1) != 1:
            pose = _capi.f32c(pose)
        if betas.dtype != torch.float32 or betas.stride(-1) != 1 or (B > 1 and betas.stride(0) == 0):
            betas = _capi.f32c(betas)
        t = self.c_tables(None if use_trans else self.center_idx, root_palm)
        verts = torch.empty(B, 778, 3, device=pose.device, dtype=torch.float32)
        joints = torch.empty(B, 21, 3, device=pose.device, dtype=torch.float32)
        flags = torch.empty(max(B, 1), device=pose.device, dtype=torch.int32) if self.check_reflection else None
        with torch.cuda.device(pose.device):
            rc = _capi.lib().dir_mano_forward(t, _capi.ptr(pose), pose.stride(0) if B > 1 else 51, _capi.ptr(betas),
                                              betas.stride(0) if B > 1 else 10, None, 0, _capi.ptr(verts),
                                              _capi.ptr(joints), None, None, _capi.ptr(flags), B,
                                              _capi.stream_ptr())
        _capi.check(rc, 'dir_mano_forward')
        if self.check_reflection and B > 0:
            # rot6d.py:50 of the reference asserts "no reflection" per sample (host sync there too)
            assert int(flags[:B].sum().item()) == 0
        return verts, joints
