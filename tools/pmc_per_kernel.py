#!/usr/bin/env python3
"""Per-kernel table from one rocprofv3 kernel trace and separate --pmc passes of the same command:
calls, average duration, HBM bytes per launch (FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE, KiB units) and the
achieved GB/s, MFMA busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) when available).
usage: pmc_per_kernel.py trace.db fetch.db write.db [sq.db]"""
import re
import sqlite3
import sys


def tables(db):
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]  # noqa: E731
    return T


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.match(r'_ZN.*?(\d+)([a-z_0-9]+_kernel)', name)
    base = re.sub(r'\(.*\)$', '', name)
    for key in ('conv_as_kernel', 'conv_pipe_kernel', 'conv_patch_kernel', 'conv_big_kernel', 'conv_igemm_kernel', 'ste_kernel', 'pgcn_layer_kernel', 'pgcn_mix_kernel',
                'mano_forward_kernel', 'grid_tokens_kernel', 'bone_fuse_kernel', 'bone_g_kernel', 'bone_vis_kernel', 'bone_proj_kernel',
                'init_head_kernel', 'regress_kernel', 'upsample_kernel', 'maxpool_kernel', 'stem_pool_kernel', 'bneck_chain_kernel', 'stem_prep_s2d', 'eval_metrics', 'gt_mano',
                'tail_chain_kernel', 'stream1x1_kernel', 'pgcn_node_kernel'):
        if key in base:
            return key
    return None


def durations(path):
    db = sqlite3.connect(path)
    T = tables(db)
    kd, ks = T('rocpd_kernel_dispatch'), T('rocpd_info_kernel_symbol')
    out = {}
    for name, s, e in db.execute('select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id=s.id' % (kd, ks)):
        k = short(name)
        if k:
            out.setdefault(k, []).append((e - s) / 1e3)
    return out


def counter(path, cname):
    db = sqlite3.connect(path)
    T = tables(db)
    pe, pi, kd, ks = T('rocpd_pmc_event'), T('rocpd_info_pmc'), T('rocpd_kernel_dispatch'), T('rocpd_info_kernel_symbol')
    q = ('select s.kernel_name, d.id, sum(e.value) from %s e join %s i on e.pmc_id=i.id join %s d on e.event_id=d.event_id '
         'join %s s on d.kernel_id=s.id where i.name=? group by d.id') % (pe, pi, kd, ks)
    out = {}
    for name, _, v in db.execute(q, (cname,)):
        k = short(name)
        if k:
            out.setdefault(k, []).append(v)
    return out


if __name__ == '__main__':
    dur = durations(sys.argv[1])
    fetch, write = counter(sys.argv[2], 'FETCH_SIZE'), counter(sys.argv[3], 'WRITE_SIZE')
    mf = bz = None
    if len(sys.argv) > 4:
        mf, bz = counter(sys.argv[4], 'SQ_VALU_MFMA_BUSY_CYCLES'), counter(sys.argv[4], 'SQ_BUSY_CU_CYCLES')
    print('%-22s %6s %10s %12s %10s %9s %s' % ('kernel', 'calls', 'avg_us', 'HBM_MB/call', 'GB/s', '%of8TB/s', 'MFMA busy (of 4 SIMDs x busy CU cycles)'))
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        n, avg = len(dur[k]), sum(dur[k]) / len(dur[k])
        f = sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [0])))
        w = sum(write.get(k, [0])) / max(1, len(write.get(k, [0])))
        mb = (2 * f + w) * 1024 / 1e6
        gbs = mb * 1e6 / (avg * 1e-6) / 1e9
        util = ''
        if mf and k in mf and bz and k in bz and sum(bz[k]) > 0:
            util = '%5.1f %%' % (100.0 * sum(mf[k]) / (4.0 * sum(bz[k])))
        print('%-22s %6d %10.1f %12.1f %10.0f %8.1f%% %s' % (k, n, avg, mb, gbs, 100 * gbs / 8000, util))
