"""Where the backward pass meets the optimiser: the gradients dir_amd/train/* returns are added into the `.grad` views of
dir_amd.optim.FlatAdamW (one contiguous fp32 buffer in parameter order = the bucket `dir_amd.dist.average_gradients` all-reduces,
SURVEY.md 8e), and one training step of a refinement stage's token half is composed from the library calls:

    stage_tokens_forward -> dir_stage_losses_backward (the 13 loss terms' gradients w.r.t. the predictions, models/dir.py:543-591)
    -> stage_tokens_backward -> add_grads -> [average_gradients] -> FlatAdamW.step            (train.py:62-70 for this part of the net)

token_stage_train_step covers the token half of one stage (image-half parameters receive no gradient there); train_step the whole network.
"""
import torch

from . import ops as O
from . import stage as TS
from .. import dist as D
from ..models import loss as L


def add_grads(named_params, prefix, grads):
    """named_params: {full name -> nn.Parameter whose .grad is a view of FlatAdamW.flat_grad}; grads: {key relative to prefix -> tensor}.
    .grad += gradient, one dir_axpy_f32 per tensor (a parameter used twice accumulates, like autograd)."""
    dsts, srcs = [], []
    for k, g in grads.items():
        p = named_params[prefix + k]
        assert p.grad is not None and p.grad.numel() == g.numel(), prefix + k
        dsts.append(p.grad)
        srcs.append(g)
    O.axpy_multi(dsts, srcs)          # one launch per 40 tensors (a shared parameter appears once per dict: its uses were summed upstream)


def inactive_parameters(named_params):
    """THE list of parameters torch leaves with `grad is None` after `sum(loss.values()).backward()` on the reference network (so
    torch.optim.AdamW neither updates nor decays them; FlatAdamW.set_inactive() reproduces that and omits them from state_dict()):
      * `backbone.fc.*`                       -- ResNet's classifier, never called (models/backbone/resnet.py:243-255 returns the pyramid);
      * `*.interaction.STEblocks.0.*`         -- the first STE block, skipped by the loop (transformer/mixSTE.py:197).
    PGraphConv's `e_0` is NOT in the list: it enters the graph through a one-entry softmax row (SemGCN/p_graph_conv.py:45-48), so torch hands
    it an identically-zero gradient tensor and AdamW still applies weight decay to it -- it stays active here too.  Works for the whole
    state dict and for any sub-tree of it (tests, tools/bench_train.py and train_step's callers all use this one function).
    With the HRNet-W48 backbone (no reference counterpart) DIR reads c2, c3, c4 only: what feeds c1 alone -- `backbone.incre.0.*` and the last
    module's `backbone.stage4.2.fuse_layers.0.*` -- has no path to the loss and is inactive too (dir_amd/train/hrnet.py forms no gradient for it)."""
    hr_dead = ('.backbone.incre.0.', '.backbone.stage4.2.fuse_layers.0.')
    return [p for k, p in named_params.items() if ('.' + k).startswith('.backbone.fc.') or '.interaction.STEblocks.0.' in '.' + k
            or ('.' + k).startswith(hr_dead)]


def token_stage_train_step(named_params, prefix, mano_tables_lr, feat_nhwc, prev, target, meta_info, faces, optimizer, coord_weight=10.0,
                           g_joint_feat=None):
    """one optimisation step of the token half of the stage whose parameters are named `prefix + key`.
    -> (stage outputs, g fusion_feat NHWC): the gradient the image half would continue from."""
    P = {k[len(prefix):]: (v.data if isinstance(v, torch.nn.Parameter) else v) for k, v in named_params.items() if k.startswith(prefix)}
    out, ctx = TS.stage_tokens_forward(P, mano_tables_lr, feat_nhwc, prev)
    cot = L.stage_loss_grads(out, target, meta_info, faces, coord_weight)
    g_feat, G = TS.stage_tokens_backward(P, mano_tables_lr, ctx, cot, g_joint_feat)
    optimizer.zero_grad()
    add_grads(named_params, prefix, G)
    D.average_gradients(optimizer.flat_grad)
    optimizer.step()
    return out, g_feat


def train_step(named_params, buffers, img, target, meta_info, faces, optimizer, overlap_allreduce=True):
    """One optimisation step of the whole network (train.py:64-70: zero_grad, forward, sum(loss).backward(), optimizer.step):
    named_params {DIR state-dict key -> nn.Parameter registered with `optimizer` (FlatAdamW)}, buffers {key -> tensor} (BatchNorm running
    statistics -- updated in place --, MANO tables).  Returns the 42 loss terms.  overlap_allreduce: bucketed gradient exchange issued
    during the backward pass (default) or one exchange after it (the round-2 path; same result, bit-identical for two ranks)."""
    from . import net as TN
    P = {k: v.data for k, v in named_params.items()}
    P.update(buffers)
    outs, ctx = TN.forward(P, img, scale_owner=optimizer)
    loss = TN.losses(outs, target, meta_info, faces)
    optimizer.zero_grad()
    if not overlap_allreduce:
        G = TN.backward(P, ctx, outs, target, meta_info, faces)
        add_grads(named_params, '', G)
        D.average_gradients(optimizer.flat_grad)
        optimizer.step()
        return loss
    # gradients move into the flat bucket as the backward pass finishes them, last layers first, and every bucket's all-reduce is issued
    # as soon as it is complete (dist.GradientBucketer): the exchange of the heads / decoder overlaps the backbone's backward
    bucketer = getattr(optimizer, '_bucketer', None)
    if bucketer is None:
        bucketer = optimizer._bucketer = D.GradientBucketer(optimizer.flat_grad, optimizer.offsets, [p.numel() for p in optimizer.params])
    index = getattr(optimizer, '_index_of', None)
    if index is None:
        index = optimizer._index_of = {id(p): i for i, p in enumerate(optimizer.params)}
    bucketer.begin()
    done = set()

    def flush(G):
        new = [k for k in G if k not in done]
        add_grads(named_params, '', {k: G[k] for k in new})
        done.update(new)
        bucketer.mark_ready([index[id(named_params[k])] for k in new])
    TN.backward(P, ctx, outs, target, meta_info, faces, flush=flush)
    bucketer.finish()
    optimizer.step()
    return loss


class GraphedTrainStep(object):
    """train_step with everything but the optimiser replayed as ONE HIP graph ("graph-captured"): zero_grad, the training-mode
    forward (with the one-launch weight packing), the objective, the backward pass and the moves into the flat gradient bucket are captured once
    the operand scales are calibrated (`warm` eager steps), then replayed; FlatAdamW.step (whose learning rate and step count are launch
    ARGUMENTS: train.py:127-149 changes them every step) and the gradient exchange of N > 1 ranks run after the replay.  The operand scales are
    arguments too, so every DIR_TRAIN_RECALIBRATE steps one step runs eagerly (re-measuring them) and the graph is captured again.
    Inputs are copied into fixed buffers; the returned loss tensors are the graph's own (valid until the next call).  Same kernels, same order:
    the parameters after n steps equal train_step's bit for bit (tests/test_gpu_full_bwd.py)."""

    def __init__(self, named_params, buffers, optimizer, faces, warm=2):
        self.named_params, self.buffers, self.optimizer, self.faces, self.warm = named_params, buffers, optimizer, faces, int(warm)
        self.graph, self.static, self.loss, self.calls, self.since_capture = None, None, None, 0, 0

    def _stage(self, img, target, meta_info):
        if self.static is None:
            dev = self.optimizer.flat_param.device
            mv = lambda t: torch.as_tensor(t).to(dev).clone()  # noqa: E731
            self.static = (mv(img), {k: mv(v) for k, v in target.items()}, {k: mv(v) for k, v in meta_info.items()})
        else:
            self.static[0].copy_(img, non_blocking=True)
            for d, src in ((self.static[1], target), (self.static[2], meta_info)):
                for k, v in d.items():
                    v.copy_(torch.as_tensor(src[k]), non_blocking=True)
        return self.static

    def _body(self):
        from . import net as TN
        img, target, meta = self.static
        opt = self.optimizer
        P = {k: v.data for k, v in self.named_params.items()}
        P.update(self.buffers)
        outs, ctx = TN.forward(P, img, scale_owner=opt)
        loss = TN.losses(outs, target, meta, self.faces)
        opt.zero_grad()
        G = TN.backward(P, ctx, outs, target, meta, self.faces)
        add_grads(self.named_params, '', G)
        return loss

    def __call__(self, img, target, meta_info):
        from . import conv as TC
        self._stage(img, target, meta_info)
        self.calls += 1
        recal = TC.RECALIBRATE > 0 and self.graph is not None and self.since_capture >= TC.RECALIBRATE
        if self.graph is None or recal:
            if recal:
                TC.reset_scales(self.optimizer)
                self.graph = None
            if self.calls <= self.warm or recal:                   # eager: measures the operand scales (host synchronisations)
                loss = self._body()
                D.average_gradients(self.optimizer.flat_grad)
                self.optimizer.step()
                self.since_capture = 0
                return loss
            pk = getattr(self.optimizer, '_dir_weight_pack', None)
            if pk is not None:
                pk.sync_table()            # scales re-measured by the eager step before this one reach the device table OUTSIDE the capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.loss = self._body()
            self.graph, self.since_capture = g, 0
        self.graph.replay()
        self.since_capture += 1
        D.average_gradients(self.optimizer.flat_grad)
        self.optimizer.step()
        return self.loss
