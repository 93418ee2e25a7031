"""CPU restatement (numpy) of the reference's training objective -- TEST INFRASTRUCTURE, never imported by the product path.

Follows models/dir.py:542-594 (the loss block of DIR.forward), models/loss.py:6-33 (NormalVectorLoss), :36-60 (EdgeLengthLoss),
:63-93 (SmoothL1Loss, knee 0.01) and models/lovasz_loss.py:19-31,155-202 (lovasz_grad / lovasz_softmax on the RAW seg logits, as the
reference calls it).  Pinned by tests/golden/g8_loss.npz = the reference's own DIR.forward in training mode (oracle/gen_golden.py
gen_loss).  Elementwise arithmetic is fp32 in the reference's operation order; reductions accumulate in float64."""
import numpy as np

F32 = np.float32
STAGE_KEYS = ('joint_left_uv', 'joint_right_uv', 'mesh_left_uv', 'mesh_right_uv', 'joint_left_xyz', 'joint_right_xyz',
              'mesh_left_xyz', 'mesh_right_xyz', 'edge_left', 'edge_right', 'normal_left', 'normal_right', 'offset')


def smooth_l1(x, y):
    """models/loss.py:68-85: per sample mean of (|z| < 0.01 ? z^2 / 2 : 0.01 (|z| - 0.005)), then the batch mean"""
    B = x.shape[0]
    z = (x.reshape(B, -1).astype(F32) - y.reshape(B, -1).astype(F32)).astype(F32)
    az = np.abs(z)
    small = az < F32(0.01)
    q = np.where(small, F32(0.5) * (z * z), F32(0.0)).astype(F32)
    l = np.where(~small, F32(0.01) * (az - F32(0.005)), F32(0.0)).astype(F32)
    per = q.astype(np.float64).mean(axis=1) + l.astype(np.float64).mean(axis=1)
    return float(per.mean())


def _normalize(v):
    n = np.sqrt((v * v).sum(axis=2, keepdims=True, dtype=F32)).astype(F32)        # F.normalize(p=2, dim=2, eps=1e-12)
    return (v / np.maximum(n, F32(1e-12))).astype(F32)


def normal_vector_loss(out, gt, face):
    """models/loss.py:11-33"""
    f0, f1, f2 = face[:, 0], face[:, 1], face[:, 2]
    v1o = _normalize(out[:, f1] - out[:, f0]); v2o = _normalize(out[:, f2] - out[:, f0]); v3o = _normalize(out[:, f2] - out[:, f1])
    v1g = _normalize(gt[:, f1] - gt[:, f0]); v2g = _normalize(gt[:, f2] - gt[:, f0])
    ng = _normalize(np.cross(v1g, v2g, axis=2).astype(F32))
    cos = [np.abs((v * ng).sum(axis=2, dtype=F32)) for v in (v1o, v2o, v3o)]
    return float(np.concatenate(cos, axis=1).astype(np.float64).mean())


def edge_length_loss(out, gt, face):
    """models/loss.py:41-60"""
    f0, f1, f2 = face[:, 0], face[:, 1], face[:, 2]

    def d(c, a, b):
        e = (c[:, a] - c[:, b]).astype(F32)
        return np.sqrt((e * e).sum(axis=2, dtype=F32) + F32(1e-12)).astype(F32)
    diffs = [np.abs(d(out, a, b) - d(gt, a, b)) for a, b in ((f0, f1), (f0, f2), (f1, f2))]
    return float(np.concatenate(diffs, axis=1).astype(np.float64).mean())


def interpolate_nearest(x, S):
    """F.interpolate(mode='nearest') (models/dir.py:565): source index floor(dst * in / out)"""
    H, W = x.shape[-2:]
    iy = np.minimum(np.floor(np.arange(S, dtype=F32) * F32(H / S)).astype(np.int64), H - 1)
    ix = np.minimum(np.floor(np.arange(S, dtype=F32) * F32(W / S)).astype(np.int64), W - 1)
    return x[..., iy[:, None], ix[None, :]]


def interpolate_bilinear(x, S):
    """F.interpolate(mode='bilinear', align_corners=False, no antialias) (models/dir.py:566): width pass inside the height pass"""
    H, W = x.shape[-2:]

    def taps(n_in):
        src = np.maximum((np.arange(S, dtype=F32) + F32(0.5)) * F32(n_in / S) - F32(0.5), F32(0.0)).astype(F32)
        i0 = np.floor(src).astype(np.int64)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = (src - i0.astype(F32)).astype(F32)
        return i0, i1, (F32(1.0) - l1).astype(F32), l1
    y0, y1, wy0, wy1 = taps(H)
    x0, x1, wx0, wx1 = taps(W)
    x = x.astype(F32)
    r0 = (wx0 * x[..., y0[:, None], x0[None, :]] + wx1 * x[..., y0[:, None], x1[None, :]]).astype(F32)
    r1 = (wx0 * x[..., y1[:, None], x0[None, :]] + wx1 * x[..., y1[:, None], x1[None, :]]).astype(F32)
    return (wy0[:, None] * r0 + wy1[:, None] * r1).astype(F32)


def cross_entropy_weighted(logits, labels, weight):
    """nn.CrossEntropyLoss(weight=w) on [B,C,S,S] logits (models/dir.py:511,567): sum_i w[y_i] nll_i / sum_i w[y_i]"""
    B, C = logits.shape[:2]
    x = logits.transpose(0, 2, 3, 1).reshape(-1, C).astype(np.float64)
    y = labels.reshape(-1)
    m = x.max(axis=1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(x - m).sum(axis=1))
    nll = lse - x[np.arange(x.shape[0]), y]
    w = np.asarray(weight, np.float64)[y]
    return float((w * nll).sum() / w.sum())


def lovasz_softmax(probas, labels):
    """models/lovasz_loss.py:155-202 with classes='present', per_image=False; `probas` are whatever the caller passes (the
    reference passes the raw logits, models/dir.py:569)"""
    B, C = probas.shape[:2]
    p = probas.transpose(0, 2, 3, 1).reshape(-1, C).astype(F32)
    y = labels.reshape(-1)
    losses = []
    for c in range(C):
        fg = (y == c).astype(F32)
        if fg.sum() == 0:
            continue
        err = np.abs(fg - p[:, c]).astype(F32)
        perm = np.argsort(-err, kind='stable')
        es, fs = err[perm].astype(np.float64), fg[perm].astype(np.float64)
        gts = fs.sum()
        inter = gts - np.cumsum(fs)
        union = gts + np.cumsum(1.0 - fs)
        jac = 1.0 - inter / union
        jac[1:] = jac[1:] - jac[:-1]
        losses.append(float((es * jac).sum()))
    return float(np.mean(losses)) if losses else 0.0


def stage_losses(pred, gt, faces, coord_weight=10.0):
    """one iteration of the loop at models/dir.py:571-592.  pred: joint_uv / mesh_uv / joint_xyz / mesh_xyz per hand + offset;
    gt: joint_2d / mesh_2d / joint_3d / mesh_3d / center per hand; faces: (left, right) int [F,3].  Returns the 13 terms."""
    out = {}
    gt_off = ((gt['center_right'] - gt['center_left']).astype(F32) / F32(0.15)).astype(F32)
    for side, face in zip(('left', 'right'), faces):
        c = gt['center_' + side].astype(F32)
        gj = ((gt['joint_3d_' + side].astype(F32) - c) / F32(0.15)).astype(F32)
        gm = ((gt['mesh_3d_' + side].astype(F32) - c) / F32(0.15)).astype(F32)
        pj = (pred['pd_joint_xyz_' + side].astype(F32) / F32(0.15)).astype(F32)
        pm = (pred['pd_mesh_xyz_' + side].astype(F32) / F32(0.15)).astype(F32)
        out['joint_%s_uv' % side] = smooth_l1(pred['pd_joint_uv_' + side], gt['joint_2d_' + side][:, :, :2]) * coord_weight
        out['mesh_%s_uv' % side] = smooth_l1(pred['pd_mesh_uv_' + side], gt['mesh_2d_' + side][:, :, :2]) * coord_weight
        out['joint_%s_xyz' % side] = smooth_l1(pj, gj) * coord_weight
        out['mesh_%s_xyz' % side] = smooth_l1(pm, gm) * coord_weight
        out['edge_' + side] = edge_length_loss(pm, gm, face)
        out['normal_' + side] = normal_vector_loss(pm, gm, face) * 0.1
    out['offset'] = smooth_l1(pred['pd_offset'], gt_off[:, 0]) * coord_weight
    return out


def dense_losses(seg_logits, dense_pred, gt_seg, gt_dense, class_weight=(0.1, 0.45, 0.45), dense_weight=1.0):
    """models/dir.py:562-569"""
    S = seg_logits.shape[-1]
    lab = interpolate_nearest(gt_seg, S).astype(np.int64)[:, 0]
    dd = interpolate_bilinear(gt_dense, S)
    return {'seg': cross_entropy_weighted(seg_logits, lab, class_weight) * 0.1 * dense_weight,
            'dense': smooth_l1(dense_pred, dd) * dense_weight,
            'lovasz': lovasz_softmax(seg_logits, lab) * 0.1 * dense_weight}


# ----------------------------------------------------------------------------- gradients (what autograd gives on the reference's modules)
def smooth_l1_grad(x, y, scale=1.0):
    """d(scale * smooth_l1(x, y)) / dx: z inside the knee, 0.01 sign(z) outside (models/loss.py:74-81), / (features * batch)"""
    B = x.shape[0]
    z = (x.reshape(B, -1).astype(F32) - y.reshape(B, -1).astype(F32)).astype(np.float64)
    g = np.where(np.abs(z) < 0.01, z, 0.01 * np.sign(z))
    return (g * (scale / (z.shape[1] * B))).reshape(x.shape)


def edge_length_grad(out, gt, face, scale=1.0):
    """d(scale * edge_length_loss) / d out"""
    out64, gt64 = out.astype(np.float64), gt.astype(np.float64)
    B, F = out.shape[0], face.shape[0]
    g = np.zeros_like(out64)
    for a, b in ((0, 1), (0, 2), (1, 2)):
        ia, ib = face[:, a], face[:, b]
        e = out64[:, ia] - out64[:, ib]
        d = np.sqrt((e * e).sum(2) + 1e-12)
        eg = gt64[:, ia] - gt64[:, ib]
        dg = np.sqrt((eg * eg).sum(2) + 1e-12)
        c = (np.sign(d - dg) / d)[..., None] * e * (scale / (B * 3 * F))
        for bi in range(B):
            np.add.at(g[bi], ia, c[bi])
            np.add.at(g[bi], ib, -c[bi])
    return g


def normal_vector_grad(out, gt, face, scale=1.0):
    """d(scale * normal_vector_loss) / d out: cos = |v_hat . n|, d cos / d e = sign(v_hat . n) (n - (v_hat . n) v_hat) / |e|"""
    out64, gt64 = out.astype(np.float64), gt.astype(np.float64)
    B, F = out.shape[0], face.shape[0]
    f0, f1, f2 = face[:, 0], face[:, 1], face[:, 2]

    def nrm(v):
        return v / np.maximum(np.sqrt((v * v).sum(2, keepdims=True)), 1e-12)
    n = nrm(np.cross(nrm(gt64[:, f1] - gt64[:, f0]), nrm(gt64[:, f2] - gt64[:, f0]), axis=2))
    g = np.zeros_like(out64)
    for hi, lo in ((f1, f0), (f2, f0), (f2, f1)):
        e = out64[:, hi] - out64[:, lo]
        ln = np.maximum(np.sqrt((e * e).sum(2, keepdims=True)), 1e-12)
        v = e / ln
        dot = (v * n).sum(2, keepdims=True)
        c = np.sign(dot) * (n - dot * v) / ln * (scale / (B * 3 * F))
        for bi in range(B):
            np.add.at(g[bi], hi, c[bi])
            np.add.at(g[bi], lo, -c[bi])
    return g


def stage_loss_grads(pred, gt, faces, coord_weight=10.0):
    """gradients of the sum of one stage's 13 terms w.r.t. pd_joint_uv / pd_mesh_uv / pd_joint_xyz / pd_mesh_xyz (per hand) and
    pd_offset (pd_mesh_uv taken as an independent input)"""
    out = {}
    gt_off = ((gt['center_right'] - gt['center_left']).astype(F32) / F32(0.15)).astype(F32)
    for side, face in zip(('left', 'right'), faces):
        c = gt['center_' + side].astype(F32)
        gj = ((gt['joint_3d_' + side].astype(F32) - c) / F32(0.15)).astype(F32)
        gm = ((gt['mesh_3d_' + side].astype(F32) - c) / F32(0.15)).astype(F32)
        pj = (pred['pd_joint_xyz_' + side].astype(F32) / F32(0.15)).astype(F32)
        pm = (pred['pd_mesh_xyz_' + side].astype(F32) / F32(0.15)).astype(F32)
        out['pd_joint_uv_' + side] = smooth_l1_grad(pred['pd_joint_uv_' + side], gt['joint_2d_' + side][:, :, :2], coord_weight)
        out['pd_mesh_uv_' + side] = smooth_l1_grad(pred['pd_mesh_uv_' + side], gt['mesh_2d_' + side][:, :, :2], coord_weight)
        out['pd_joint_xyz_' + side] = smooth_l1_grad(pj, gj, coord_weight) / 0.15
        out['pd_mesh_xyz_' + side] = (smooth_l1_grad(pm, gm, coord_weight) + edge_length_grad(pm, gm, face)
                                      + normal_vector_grad(pm, gm, face, 0.1)) / 0.15
    out['pd_offset'] = smooth_l1_grad(pred['pd_offset'], gt_off[:, 0], coord_weight)
    return out


def dense_loss_grads(seg_logits, dense_pred, gt_seg, gt_dense, class_weight=(0.1, 0.45, 0.45), dense_weight=1.0):
    """gradients of seg + dense + lovasz w.r.t. the seg logits and the dense prediction"""
    B, C, S, _ = seg_logits.shape
    lab = interpolate_nearest(gt_seg, S).astype(np.int64)[:, 0]
    dd = interpolate_bilinear(gt_dense, S)
    x = seg_logits.transpose(0, 2, 3, 1).reshape(-1, C).astype(np.float64)
    y = lab.reshape(-1)
    p = np.exp(x - x.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
    w = np.asarray(class_weight, np.float64)[y]
    gce = p.copy(); gce[np.arange(len(y)), y] -= 1.0
    gce *= (w / w.sum())[:, None] * 0.1 * dense_weight
    glov = np.zeros_like(x)
    present = [c for c in range(C) if (y == c).any()]
    for c in present:
        fg = (y == c).astype(np.float64)
        err = np.abs(fg - x[:, c].astype(F32).astype(np.float64))
        perm = np.argsort(-err.astype(F32), kind='stable')
        fs = fg[perm]
        gts = fs.sum()
        jac = 1.0 - (gts - np.cumsum(fs)) / (gts + np.cumsum(1.0 - fs))
        jac[1:] = jac[1:] - jac[:-1]
        glov[perm, c] = np.sign(x[perm, c] - fs) * jac * (0.1 * dense_weight / len(present))
    gseg = (gce + glov).reshape(B, S, S, C).transpose(0, 3, 1, 2)
    return {'seg': gseg, 'dense': smooth_l1_grad(dense_pred, dd, dense_weight)}
