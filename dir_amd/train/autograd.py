"""torch.autograd bridges for the stand-alone mirror modules in TRAINING mode.

The reference's operator classes are trainable `nn.Module`s (SemGCN/p_gcn.py:20-27, SemGCN/p_graph_conv.py:39-59, transformer/mixSTE.py:194-205,
manopth/manopth/manolayer.py:110-270: torch autograd differentiates whatever `train.py:64-70` builds from them).  The mirrors' eval-mode
forwards are single fused launches without an autograd graph; in `.train()` mode they run through the functions of dir_amd/train/ (the kernels
of the whole-network training step: exact-fp32 GEMMs, batch-statistics BatchNorm with running-statistics update, LayerNorm / attention / GELU
backward, dir_mano_backward_pair) wrapped in ONE autograd node per module call: parameters receive `.grad` like under the reference, inputs
receive their gradient, and everything between is library calls.  Plumbing only -- no arithmetic here.
"""
import torch


class _Node(torch.autograd.Function):
    """forward_fn(P, *inputs) -> (tuple of outputs, saved); backward_fn(P, saved, *grad_outputs) -> (tuple of input gradients, {name: gradient})"""

    @staticmethod
    def forward(ctx, forward_fn, backward_fn, names, buffers, n_in, dirty_first, *tensors):
        inputs, params = tensors[:n_in], tensors[n_in:]
        P = {k: v.detach() for k, v in zip(names, params)}
        P.update(buffers)
        outs, saved = forward_fn(P, *[t.detach() for t in inputs])
        ctx.P, ctx.saved_ctx, ctx.backward_fn, ctx.names, ctx.n_in = P, saved, backward_fn, names, n_in
        ctx.dirty_first = dirty_first
        ctx.set_materialize_grads(False)        # an unused output's gradient stays None (no zero tensors made, none added)
        if dirty_first:                       # the module updates its first input IN PLACE like the reference does (STE: x += pos, mixSTE.py:196):
            ctx.mark_dirty(tensors[0])        # autograd then tracks later uses of that tensor through this node
            return (tensors[0],) + tuple(outs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        g_dirty = None
        if ctx.dirty_first:
            g_dirty, gouts = gouts[0], gouts[1:]
        gouts = [None if g is None else g.contiguous() for g in gouts]
        if ctx.dirty_first:                   # (the gradient reaching later uses of the updated first input: the module's backward adds it where it belongs)
            g_in, G = ctx.backward_fn(ctx.P, ctx.saved_ctx, *gouts, g_dirty=None if g_dirty is None else g_dirty.contiguous())
        else:
            g_in, G = ctx.backward_fn(ctx.P, ctx.saved_ctx, *gouts)
        grads = []
        for k in ctx.names:
            g = G.get(k)
            grads.append(None if g is None else g.clone())          # a fresh tensor: autograd may adopt it as .grad (train/pgcn.py shares its e_0 zeros)
        return (None, None, None, None, None, None) + tuple(g_in) + tuple(grads)


def run(forward_fn, backward_fn, inputs, params, buffers=None, dirty_first=False):
    """params: {name: nn.Parameter}; buffers: {name: tensor} (running statistics: updated in place by the forward).  Returns the outputs tuple."""
    names = tuple(params.keys())
    outs = _Node.apply(forward_fn, backward_fn, names, dict(buffers or {}), len(inputs), dirty_first, *inputs, *[params[k] for k in names])
    return outs[1:] if dirty_first else outs
