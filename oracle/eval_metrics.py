"""TEST INFRASTRUCTURE (CPU oracle, numpy) -- never imported by the product path.

Restatement of the evaluation-metric maths of the reference's apps/eval.py:
  * Jr            apps/eval.py:22-44   (16-row MANO joint regressor + 5 one-hot fingertip rows, re-ordered to 21)
  * xyz2uvd       apps/eval.py:82-85   (pin-hole projection with the per-sample 3x3 intrinsics)
  * batch_metrics apps/eval.py:151-241 (GT joints re-regressed from GT vertices, root + bone-length-scale alignment,
                                        per-joint / per-vertex 3-D and 2-D L2 errors, relative-root error)
  * summarize     apps/eval.py:246-306 (concatenate over batches, means, mm scaling)
Pinned by tests/golden/g9_eval.npz, which oracle/gen_golden.py::gen_eval produced by EXECUTING the reference's own
loop body (extracted from the file at generation time) on the synthetic batch of gen_golden.eval_inputs.
"""
import numpy as np

TIP_VERTS = (745, 317, 444, 556, 673)                      # apps/eval.py:29-33
JR_ORDER = (0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20)   # apps/eval.py:35-39
F = np.float32


def jr_matrix(j_regressor):
    """apps/eval.py:26-40: [16,778] -> [21,778]"""
    j = np.asarray(j_regressor, F)
    tips = np.zeros((5, j.shape[1]), F)
    for r, v in enumerate(TIP_VERTS):
        tips[r, v] = 1.0
    return np.ascontiguousarray(np.concatenate([j, tips], 0)[list(JR_ORDER)])


def xyz2uvd(p, cam):
    """apps/eval.py:82-85"""
    q = np.matmul(p, cam.transpose(0, 2, 1))
    return q[:, :, :2] / q[:, :, 2:]


def _norm(x):
    return np.sqrt((x * x).sum(-1, dtype=x.dtype)).astype(x.dtype)


def batch_metrics(ins, root_joint=0, use_scale=True, dtype=F):
    """apps/eval.py:151-241 for one batch.  `ins`: verts_pd_{left,right} [B,778,3] (result[-1]['pd_mesh_xyz_*']),
    pd_offset [B,3], verts_gt_* [B,778,3] (data[3], data[5]), verts2d_gt_* [B,778,2] (data[7], data[9]), cam [B,3,3]
    (data[10]), jr_* [21,778] (jr_matrix of each hand's regressor)."""
    out, keep = {}, {}
    F = dtype          # float32 = the reference's arithmetic; float64 = the exact value for arbitration tests
    ins = {k: np.asarray(v, F) for k, v in ins.items()}
    for side in ('left', 'right'):
        jr, cam = ins['jr_' + side], ins['cam']
        vg, vp = ins['verts_gt_' + side].astype(F), ins['verts_pd_' + side].astype(F)
        jg = np.matmul(jr, vg)                                            # :151-152
        j2g = xyz2uvd(jg, cam)                                            # :153-154
        root_g = jg[:, root_joint:root_joint + 1].copy()                  # :157-158
        len_g = _norm(jg[:, 9] - jg[:, 0])                                # :160-161
        jg_rel, vg_rel = jg - root_g, vg - root_g                         # :162-165
        jp_ori = np.matmul(jr, vp)                                        # :173-174
        root_p = jp_ori[:, root_joint:root_joint + 1].copy()              # :176-177
        len_p = _norm(jp_ori[:, 9] - jp_ori[:, 0])                        # :178-179
        scl = (len_g / len_p)[:, None, None] if use_scale else F(1)       # :180-185
        jp, vp_al = (jp_ori - root_p) * scl, (vp - root_p) * scl          # :187-190
        out['joint_err_' + side] = _norm(jp - jg_rel)                     # :192
        out['vert_err_' + side] = _norm(vp_al - vg_rel)                   # :204
        out['joints_pd_' + side], out['joints_gt_' + side] = jp, jg_rel   # :195-196
        out['vert2d_err_' + side] = _norm(xyz2uvd(vp_al + root_g, cam) - ins['verts2d_gt_' + side])   # :212,217
        out['joint2d_err_' + side] = _norm(xyz2uvd(jp + root_g, cam) - j2g)                          # :214,225
        keep[side] = (jg, jp_ori)
    gt_off = keep['right'][0][:, root_joint] - keep['left'][0][:, root_joint]                        # :156
    rel = ins['pd_offset'].astype(F) * F(0.15)                                                       # :170
    if root_joint != 0:                                                                              # :236-238
        rel = (keep['right'][1][:, root_joint] + rel) - keep['left'][1][:, root_joint]
    out['root_err'] = _norm(gt_off - rel)[:, None]                                                   # :234,239
    return out


def summarize(batches):
    """apps/eval.py:246-306: concatenate the per-batch arrays and reduce to the printed numbers."""
    cat = {k: np.concatenate([b[k] for b in batches], 0) for k in batches[0]}
    s = {}
    for nm, key, mul in (('joint_mm', 'joint_err_', 1000), ('vert_mm', 'vert_err_', 1000),
                         ('joint_px', 'joint2d_err_', 1), ('vert_px', 'vert2d_err_', 1)):
        l, r = cat[key + 'left'].mean() * mul, cat[key + 'right'].mean() * mul
        s[nm] = {'left': float(l), 'right': float(r), 'all': float((l + r) / 2)}
    s['root_mm'] = float(cat['root_err'].reshape(-1).mean() * 1000)
    return s
