"""Drop-in for the reference's models/dir.py (DIR and its sub-modules): identical constructor signatures, parameter
tree and state-dict keys (963 keys; a reference checkpoint's ['net'] loads with strict=True), identical forward
signature and output structure (models/dir.py:513-540).  The classes below are parameter containers; DIR.forward runs
the whole eval-mode path through dir_amd.engine.DirEngine, i.e. through libdir_hip.so (include/dir_hip.h).

Scope: eval mode = the benchmarked hot path (DirEngine); `self.training == True` runs the fp32 training forward of
dir_amd/train/net.py and returns the 42 loss terms attached to the parameters (row 8f.2: `sum(loss.values()).backward()` fills `.grad`).
`compute_dtype` selects bf16 feature maps (BASELINE config 2, default), exact-fp32 MFMA convs, or the f16x3 split-precision mode.
"""
import torch
import torch.nn as nn

from .. import _capi
from ..engine import DirEngine
from ..manopth.manolayer import ManoLayer as ObmanManoLayer
from ..SemGCN.p_gcn import ResSimplePGCN
from ..SemGCN.utils import adj_mx_from_edges, get_sketch_setting
from ..transformer.mixSTE import STE
from .backbone.hourglass import Residual
from .backbone.resnet import resnet50 as ResNet50


def _token_mlp(cin, cmid, cout):
    return nn.Sequential(nn.Conv1d(cin, cmid, 1), nn.BatchNorm1d(cmid), nn.ReLU(), nn.Conv1d(cmid, cout, 1))


def _mano_pair(mano_path, root_joint):
    kw = dict(root_rot_mode='6D', joint_rot_mode='axisang', use_pca=True, mano_root=mano_path, ncomps=45,
              center_idx=root_joint, flat_hand_mean=False, robust_rot=True)
    return ObmanManoLayer(side='right', **kw), ObmanManoLayer(side='left', **kw)


def _fix_shape(mano_layer_left, mano_layer_right):
    # models/dir.py:306-309
    if torch.sum(torch.abs(mano_layer_left.th_shapedirs[:, 0, :] - mano_layer_right.th_shapedirs[:, 0, :])) < 1:
        print('Fix shapedirs bug of MANO')
        mano_layer_left.th_shapedirs[:, 0, :] *= -1


class ImgFeature2JointFeature(nn.Module):
    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.filters = _token_mlp(in_dim, out_dim, out_dim)


class RegressorOffset(nn.Module):
    def __init__(self, feat_dim, mano_path, root_joint):
        super().__init__()
        self.mano_layer_right, self.mano_layer_left = _mano_pair(mano_path, root_joint)
        _fix_shape(self.mano_layer_left, self.mano_layer_right)
        d = 3 * 2 + 15 * 3 + 10 + 3
        self.mano_left = nn.Linear(feat_dim + d, d)
        self.mano_right = nn.Linear(feat_dim + d, d)
        self.offset = nn.Linear(feat_dim * 2 + 3, 3)
        for m in (self.mano_left, self.mano_right, self.offset):
            m.weight.data.normal_(0, 0.001)


class Joint2BoneFeature(nn.Module):
    def __init__(self, img_feat_dim, emd_dim, joint_dim, joint_num, feature_size, mano_pth, root_joint, distance=1):
        super().__init__()
        adj = adj_mx_from_edges(joint_num, get_sketch_setting(), sparse=False, eye=False)
        self.bone_num = 20
        self.parent = torch.Tensor([0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 0, 13, 14, 15, 0, 17, 18, 19]).long()
        self.child = torch.Tensor([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20]).long()
        self.gcn_left = ResSimplePGCN(adj, emd_dim, num_layers=4)
        self.gcn_right = ResSimplePGCN(adj, emd_dim, num_layers=4)
        self.img2joint_left = ImgFeature2JointFeature(img_feat_dim, emd_dim)
        self.img2joint_right = ImgFeature2JointFeature(img_feat_dim, emd_dim)
        self.pos_emb_left = _token_mlp(3, emd_dim, emd_dim)
        self.pos_emb_right = _token_mlp(3, emd_dim, emd_dim)
        self.global_pos_emb = _token_mlp(3, emd_dim, emd_dim)
        self.interaction = STE(num_joints=joint_num * 2, in_chans=emd_dim, out_dim=joint_dim, depth=4)
        self.proj_feat_emb = _token_mlp(joint_dim, joint_dim, joint_dim)
        self.fusion = nn.Sequential(nn.Conv2d(joint_dim * self.bone_num * 2, img_feat_dim, 3, 1, 1),
                                    nn.BatchNorm2d(img_feat_dim), nn.ReLU(), nn.Conv2d(img_feat_dim, img_feat_dim, 1))
        self.regressor = RegressorOffset(joint_num * joint_dim, mano_pth, root_joint)
        c = torch.arange(feature_size) + 0.5
        gx, gy = torch.meshgrid(c, c, indexing='ij')
        self.register_buffer('img_gird', torch.stack((gy, gx), dim=-1).reshape([feature_size ** 2, 2]).contiguous())
        self.joint_dim, self.joint_num, self.feature_size, self.distance = joint_dim, joint_num, feature_size, distance

    def bone_proj(self, joint_uv, joint_feat):
        """models/dir.py:146-174 for ONE hand: [B,21,2], [B,21,64] -> [B,1280,S,S] (float32)."""
        _capi.require_cuda(joint_uv, joint_feat)
        B, S = joint_feat.shape[0], self.feature_size
        uv = _capi.f32c(joint_uv.detach())
        emb = torch.cat([joint_feat.detach().float(), joint_feat.detach().float()], 1).contiguous()
        out = torch.empty(B, S, S, 2560, device=uv.device)
        with torch.cuda.device(uv.device):
            _capi.check(_capi.lib().dir_bone_proj_forward(_capi.ptr(uv), _capi.ptr(uv), _capi.ptr(emb), _capi.ptr(out),
                                                          None, None, B, S, float(self.distance), 0, _capi.stream_ptr()),
                        'dir_bone_proj_forward')
        return out[..., :1280].permute(0, 3, 1, 2)


class InitRegressor(nn.Module):
    def __init__(self, feat_dim, mano_path, root_joint):
        super().__init__()
        self.mano_layer_right, self.mano_layer_left = _mano_pair(mano_path, root_joint)
        _fix_shape(self.mano_layer_left, self.mano_layer_right)

        def attn():
            return nn.Sequential(nn.Conv2d(feat_dim, feat_dim // 2, 3, 1, 1), nn.BatchNorm2d(feat_dim // 2), nn.ReLU(),
                                 nn.Conv2d(feat_dim // 2, 1, 1, 1), nn.Sigmoid())
        self.attention_left, self.attention_right = attn(), attn()
        self.offset = nn.Linear(feat_dim, 3)
        self.mano_left = nn.Linear(feat_dim, 64)
        self.mano_right = nn.Linear(feat_dim, 64)
        for m in (self.mano_left, self.mano_right, self.offset):
            m.weight.data.normal_(0, 0.001)


class FusionJointInterIterDecoder(nn.Module):
    def __init__(self, joint_num, mano_pth, root_joint, inDim=[2048, 1024, 512, 256], fDim=[256, 256, 256, 256], extra_stages=0):
        super().__init__()
        self.up4 = nn.Upsample(scale_factor=2, mode='bilinear')
        self.skip_layer4 = Residual(inDim[1], fDim[0])
        self.fusion_layer4 = Residual(inDim[0] + fDim[0], fDim[1])
        self.projecter_4 = Joint2BoneFeature(fDim[1], 128, 64, joint_num, 16, mano_pth, root_joint, distance=1)
        self.enhance_layer4 = Residual(fDim[1] * 2, fDim[1])
        self.up3 = nn.Upsample(scale_factor=2, mode='bilinear')
        self.skip_layer3 = Residual(inDim[2], fDim[1])
        self.fusion_layer3 = Residual(fDim[1] * 2, fDim[2])
        self.projecter_3 = Joint2BoneFeature(fDim[2], 128, 64, joint_num, 32, mano_pth, root_joint, distance=2)
        self.enhance_layer3 = Residual(fDim[2] * 2, fDim[2])
        self.conv_final = nn.Sequential(nn.Conv2d(fDim[3], fDim[3], 3, 1, 1, bias=False), nn.BatchNorm2d(fDim[3]),
                                        nn.ReLU(True), nn.Conv2d(fDim[3], fDim[3], 1, 1))

        def head():
            return nn.Sequential(nn.Conv2d(fDim[3], fDim[3] // 2, 3, 1, 1), nn.BatchNorm2d(fDim[3] // 2), nn.ReLU(),
                                 nn.Conv2d(fDim[3] // 2, 3, 1, 1))
        self.seg, self.dense = head(), head()
        # f4 (no reference counterpart): N further refinement iterations at the final 32x32 resolution, each with its
        # own Joint2BoneFeature (like projecter_3) and Residual 512 -> 256 (like enhance_layer3); keys decoder.projecter_x.<i>.* /
        # decoder.enhance_layer_x.<i>.*.  With 0 (the default) the module tree and its 963 keys are exactly the reference's.
        if extra_stages:
            self.projecter_x = nn.ModuleList(Joint2BoneFeature(fDim[2], 128, 64, joint_num, 32, mano_pth, root_joint, distance=2) for _ in range(extra_stages))
            self.enhance_layer_x = nn.ModuleList(Residual(fDim[2] * 2, fDim[2]) for _ in range(extra_stages))


class _TrainObjective(torch.autograd.Function):
    """the whole training-mode forward + objective as ONE autograd node over the parameters: forward = dir_amd.train.net.forward + losses,
    backward = dir_amd.train.net.backward with the upstream gradient of each of the 42 terms"""
    @staticmethod
    def forward(ctx, box, img, target, meta_info, faces, buffers, keys, *params):
        from ..train import net as TN
        P = {k: p.detach() for k, p in zip(keys, params)}
        P.update(buffers)
        outs, c = TN.forward(P, img, scale_owner=box.get('owner'))
        loss = TN.losses(outs, target, meta_info, faces)
        ctx.saved = (P, c, outs, target, meta_info, faces, keys, list(loss))
        box['outs'], box['keys'] = outs, list(loss)
        return torch.stack([loss[k].reshape(()) for k in loss])

    @staticmethod
    def backward(ctx, g_vec):
        from ..train import net as TN
        P, c, outs, target, meta_info, faces, keys, lkeys = ctx.saved
        G = TN.backward(P, c, outs, target, meta_info, faces, grad_out={k: g_vec[i] for i, k in enumerate(lkeys)})
        return (None,) * 7 + tuple(G.get(k) for k in keys)


class DIR(nn.Module):
    def __init__(self, joint_num, mano_path, root_joint=0, compute_dtype=torch.bfloat16, extra_stages=0, arith=None, backbone='resnet50'):
        """extra_stages / arith are this build's extensions (defaults = the reference's network): extra_stages = N more refinement iterations at
        32x32 (config 5's "5 refinement iters" = extra_stages 2; outs_list then carries 3 + N stage dicts before the dense / seg dict; trains
        like the rest: 3 + 13 (3 + N) loss terms); arith = 'f16x3' with compute_dtype float32: the split-precision parity mode (DirEngine)."""
        super().__init__()
        self.extra_stages, self.arith = int(extra_stages), arith
        self.joint_num = joint_num
        self.root_joint = root_joint
        self.compute_dtype = compute_dtype
        assert backbone in ('resnet50', 'hrnet_w48')
        if backbone == 'hrnet_w48':       # BASELINE config 5; no reference counterpart (dir_amd/models/backbone/hrnet.py)
            from .backbone.hrnet import hrnet_w48
            self.backbone = hrnet_w48()
        else:
            self.backbone = ResNet50()    # ImageNet weights are a download in the reference (models/dir.py:490-498)
        self.backbone_name = backbone
        self.mesh_sample_num = joint_num
        self.init_regressor = InitRegressor(self.backbone.inplanes, mano_path, root_joint)
        self.decoder = FusionJointInterIterDecoder(self.joint_num, mano_path, root_joint, extra_stages=self.extra_stages)
        self.coord_weight, self.dense_weight = 10, 1
        self.seg_loss = nn.CrossEntropyLoss(weight=torch.Tensor([.1, 0.45, 0.45]))     # state-dict key seg_loss.weight
        self._engine, self._engine_key, self._sd_tensors = None, None, None
        self.autotune = True

    def _tensors(self):
        """parameters and buffers in state-dict order.  What is cached is the list of SLOTS -- (the owning module's _parameters / _buffers
        dict, name) -- not the tensor objects: every call looks the current tensor of each slot up again, so `model.backbone.float()`,
        `.to()` on a sub-module or `bn.running_mean = t` (which replace tensor objects without going through DIR._apply) are seen.  The
        slot list itself is rebuilt after _apply / load_state_dict / refresh()."""
        if self._sd_tensors is None:
            slots = []
            for mod in self.modules():
                slots += [(mod._parameters, k) for k, v in mod._parameters.items() if v is not None]
                slots += [(mod._buffers, k) for k, v in mod._buffers.items() if v is not None and k not in mod._non_persistent_buffers_set]
            self._sd_tensors = slots
        return [d[k] for d, k in self._sd_tensors]

    def refresh(self):
        """forget the packed engine: the next forward re-packs the parameters (needed only after out-of-band edits that bump neither
        a tensor's version counter nor its storage, e.g. writes through a raw pointer)"""
        self._sd_tensors, self._engine_key = None, None

    def _apply(self, fn, *args, **kw):               # .cuda() / .to() / .float(): storages change
        self._sd_tensors = None
        return super()._apply(fn, *args, **kw)

    def load_state_dict(self, *args, **kw):
        self._sd_tensors = None
        return super().load_state_dict(*args, **kw)

    def engine(self):
        """(re)pack the parameters when any of them changed (load_state_dict, .to(), in-place edits, optimiser steps)."""
        key = (self.compute_dtype, self.arith) + tuple((t.data_ptr(), t._version) for t in self._tensors())
        if key != self._engine_key:
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            dev = next(self.parameters()).device
            if dev.type != 'cuda':
                raise _capi.DirHipError('DIR runs on the GPU only: call .cuda() first (no CPU fallback exists)')
            tuned = self._engine.export_all_tuning() if self._engine is not None else None
            self._engine = DirEngine(sd, dtype=self.compute_dtype, root_joint=self.root_joint, device=dev, arith=self.arith)
            if tuned:
                self._engine.pending_tuning = tuned      # same architecture, new weights: the kernel choices carry over
            self._engine_key = key
        return self._engine

    def objective(self, outs_list, target, meta_info):
        """The loss block of the reference's forward (models/dir.py:542-594) evaluated on eval-mode outputs: the same 42 keys
        and weights (coord_weight 10, dense_weight 1, seg class weights .1/.45/.45), forward values only (validation loss)."""
        from .loss import DirLoss
        crit = DirLoss(self.init_regressor.mano_layer_left.th_faces, self.init_regressor.mano_layer_right.th_faces)
        return crit(outs_list[:-1], outs_list[-1], target, meta_info)

    def _forward_train(self, input, target, meta_info):
        """Training mode (train.py:66-68 runs `outs_list, loss = model(inputs, targets, meta_infos); sum(loss[k] ...).backward()`): the
        fp32 training forward of dir_amd/train/net.py (batch-statistics BatchNorm, running statistics updated) and the 42 loss terms as
        0-d tensors attached to the parameters through one autograd node whose backward is dir_amd.train.net.backward -- `.backward()`
        on any combination of the terms fills `.grad` of every trained parameter like the reference does."""
        from ..train import net as TN
        x = _capi.f32c(input['img'].cuda())
        named = [(k, p) for k, p in self.named_parameters()]
        buffers = {k: b for k, b in self.named_buffers() if 'num_batches_tracked' not in k}
        faces = (self.init_regressor.mano_layer_left.th_faces, self.init_regressor.mano_layer_right.th_faces)
        box = {'owner': self}
        vec = _TrainObjective.apply(box, x, target, meta_info, faces, buffers, [k for k, _ in named], *[p for _, p in named])
        with torch.no_grad():
            for k, b in self.named_buffers():
                if k.endswith('num_batches_tracked'):               # nn.BatchNorm bookkeeping (momentum is not None: unused by the maths):
                    # one count per BatchNorm CALL -- global_pos_emb and proj_feat_emb are shared modules the reference calls once per
                    # hand (models/dir.py:106-107,118-119), so theirs advance by 2 per forward
                    b += 2 if ('.global_pos_emb.' in k or '.proj_feat_emb.' in k) else 1
        loss = {k: vec[i] for i, k in enumerate(box['keys'])}
        outs = box['outs']
        outs_list = [{k: o.get(k) for k in ('pd_joint_uv_left', 'pd_joint_uv_right', 'pd_mesh_xyz_left', 'pd_mesh_xyz_right', 'pd_joint_xyz_left',
                                             'pd_joint_xyz_right', 'pd_offset')} for o in outs[:-1]]
        for o, d in zip(outs[:-1], outs_list):
            d['pd_proj_left'], d['pd_proj_right'], d['pd_rel_joint'] = o['pd_mano_para_left'][:, 61:], o['pd_mano_para_right'][:, 61:], None
        outs_list.append({'dense': outs[-1]['dense'], 'seg': outs[-1]['seg'], 'proj_feat': outs[-1].get('proj_feat')})    # models/dir.py:536-540
        return outs_list, loss

    def forward(self, input, target, meta_info):
        if self.training:
            return self._forward_train(input, target, meta_info)      # either backbone (HRNet-W48 since round 5: dir_amd/train/hrnet.py)
        x = input['img'].cuda()                                   # the reference moves the input itself (models/dir.py:514)
        eng = self.engine()
        with torch.cuda.device(x.device), torch.no_grad():
            x = x.contiguous() if x.dtype == torch.uint8 else _capi.f32c(x)       # uint8 BGR [B,256,256,3]: fused normalisation
            if not eng.calibrated and eng.arith is not None and not torch.cuda.is_current_stream_capturing():
                eng.calibrate(x)          # f16x3 / f16: per-layer power-of-two input scales from the FIRST batch this engine sees (64x headroom; a later
                #                           batch with much larger activations saturates at the f16 maximum: call self.engine().calibrate(batch) again)
            if self.autotune and x.shape[0] not in eng.tuned_batches and not torch.cuda.is_current_stream_capturing():
                eng.tune_for(x)             # per-layer conv kernel choice (bit-identical results): timed once, re-used for other batch sizes
            flags = torch.zeros(3 + self.extra_stages, 2, x.shape[0], device=x.device, dtype=torch.int32)
            outs = eng.forward(x, reflection_flags=flags)
            # manopth's robust 6D -> rotation asserts det > 0 over the batch (rot6d.py:50, a host synchronisation in the reference
            # too); a stage whose predicted root rotation is a reflection raises exactly there
            if not torch.cuda.is_current_stream_capturing() and int(flags.sum().item()) != 0:
                raise AssertionError('robust_compute_rotation_matrix_from_ortho6d: det < 0 (rot6d.py:50) in stage(s) %s'
                                     % sorted(set(torch.nonzero(flags.sum((1, 2))).flatten().tolist())))
        outs_list = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()} for o in outs]
        return outs_list, {}
