#!/usr/bin/env python3
"""Average HBM traffic per launch of the dominant kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
the bench command.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes of a wide
coalesced read -> doubled; both counters are in KiB.  Writes profiles/<tag>_pmc_traffic.json (read by bench.py)."""
import json, os, sqlite3, sys


def per_kernel(path, counter, pat):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]  # noqa: E731
    pe, pi, kd, ks = T('rocpd_pmc_event'), T('rocpd_info_pmc'), T('rocpd_kernel_dispatch'), T('rocpd_info_kernel_symbol')
    q = ("select d.id, sum(e.value) from %s e join %s i on e.pmc_id=i.id join %s d on e.event_id=d.event_id "
         "join %s s on d.kernel_id=s.id where (%s) and i.name=? group by d.id") % (
             pe, pi, kd, ks, ' or '.join('s.kernel_name like ?' for _ in pat))
    vals = [v for _, v in db.execute(q, tuple('%' + x + '%' for x in pat) + (counter,))]
    return len(vals), sum(vals)


if __name__ == '__main__':
    fetch_db, write_db, out = sys.argv[1:4]
    # the 16-bit -> 16-bit convolution family (bf16: `t`, f16 storage: `f16s_t`; the command runs ONE of them: --no-other-half): conv.hip (4-wave)
    # and conv_pipe.hip (8-wave pipelined / halo-reuse) kernels, the chains and the streaming 1x1
    h = 'f16s_tE'
    pat = ['conv_igemm_kernelItt', 'conv_igemm_kernelIN3dir5convk6' + h + 'S3_', 'conv_pipe_kernelIt', 'conv_pipe_kernelINS0_6' + h, 'conv_patch_kernelIt',
           'conv_patch_kernelINS0_6' + h, 'conv_big_kernelIt', 'conv_big_kernelINS0_6' + h, 'bneck_chain_kernel', 'tail_chain_kernel', 'stream1x1_kernel', 'conv_as_kernel']
    nf, f = per_kernel(fetch_db, 'FETCH_SIZE', pat)
    nw, w = per_kernel(write_db, 'WRITE_SIZE', pat)
    res = {'kernel': 'conv family (conv_igemm / conv_pipe / conv_patch / conv_big / bneck_chain / tail_chain / stream1x1 / conv_as; 16-bit storage)', 'launches_fetch_pass': nf, 'launches_write_pass': nw,
           'fetch_kib_per_launch_raw': f / nf, 'write_kib_per_launch_raw': w / nw,
           'hbm_bytes_per_launch': (2.0 * f / nf + w / nw) * 1024.0,
           'head': os.environ.get('DIR_HEAD', 'unrecorded'),     # git HEAD the counters were taken at (the GPU box has no .git: passed in)
           'note': 'FETCH_SIZE doubled (gfx950: counter tallies 128-B requests at 64 B), WRITE_SIZE as reported; '
                   'separate --pmc passes of `python bench.py --no-graph --steps 4 --warmup 2 --no-other-half --no-pgcn ...` (tools/profile_round.sh)'}
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res))
