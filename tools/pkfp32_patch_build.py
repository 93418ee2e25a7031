"""Investigation aid for "Packed FP32 beside another kernel" (DESIGN.md): rebuilds csrc/mano.hip WITH packed-FP32 instructions, edits the device
assembly of mano_forward_kernel<256, 1> (the victim of tools/pkfp32_repro.hip) with one transform, and links mano + core into
tools/_ubench/libmano_pk_<name>.so for DIR_VICTIM_LIB.  Runs on CPU (hipcc cross-compiles).  python tools/pkfp32_patch_build.py [name ...]
  nop_before_pk     s_nop 4 before every v_pk_{fma,mul,add}_f32            (a VALU / operand hazard the hardware does not interlock)
  lgkm_before_pk    s_waitcnt lgkmcnt(0) before every v_pk_*_f32            (an LDS / scalar-load result consumed too early)
  vm_before_pk      s_waitcnt vmcnt(0) before every v_pk_*_f32              (a global-load result consumed too early)
  nop_after_ds      s_nop 4 after every ds_read*
  first_half / second_half   nop_before_pk + lgkm_before_pk on the first / second half of the kernel's packed instructions only
  depack            every v_pk_{fma,mul,add}_f32 of the kernel replaced IN THE ASSEMBLY by its two scalar halves (v_fma_f32 / v_mul_f32 / v_add_f32 with
                    the same op_sel / neg selections, low result through a spare VGPR): registers, memory instructions and schedule otherwise
                    unchanged -- separates "the packed instruction executes wrongly" from "something else in the packed build"
Never part of the product build."""
import os, re, shlex, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORK = '/tmp/pkpatch'
KERNEL = '_ZN12_GLOBAL__N_119mano_forward_kernelILi256ELi1EEEvNS_8ManoArgsE'
PK = re.compile(r'^\s*v_pk_(fma|mul|add)_f32\b')


def run(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    if r.returncode != 0:
        sys.exit('failed: %s\n%s' % (cmd if isinstance(cmd, str) else ' '.join(cmd), r.stderr[-3000:]))
    return r


# v_pk_fma_f32 forms of the kernel by which half of which source they broadcast (op_sel = half feeding the LOW result, op_sel_hi = the HIGH one)
PATTERNS = {'b1lo': r'op_sel_hi:\[1,0,1\]\s*$', 'b1hi': r'op_sel:\[0,1,0\]\s*$', 'b0lo': r'op_sel_hi:\[0,1,1\]', 'b2lo': r'op_sel_hi:\[1,1,0\]',
            'b12lo': r'op_sel_hi:\[1,0,0\]', 'b1hi_lo': r'op_sel:\[0,1,0\] op_sel_hi:\[1,0,1\]', 'b1hiany': r'op_sel:\[0,1,0\]',
            'b1any': r'op_sel:\[0,1,0\]|op_sel_hi:\[1,0,1\]'}
PKFULL = re.compile(r'^\s*v_pk_(fma|mul|add)_f32\s+(.*)$')


def _halves(tok):
    m = re.match(r'^([vs])\[(\d+):(\d+)\]$', tok)
    if m:
        return '%s%s' % (m.group(1), m.group(2)), '%s%s' % (m.group(1), m.group(3))
    return tok, tok                      # an inline constant feeds both halves


def depack_line(ln, tmp):
    m = PKFULL.match(ln)
    op, rest = m.group(1), m.group(2).split(';')[0].strip()
    mods = dict(op_sel=None, op_sel_hi=None, neg_lo=None, neg_hi=None)
    for k in list(mods):
        mm = re.search(r'\b%s:\[([0-9,]+)\]' % k, rest)
        if mm:
            mods[k] = [int(x) for x in mm.group(1).split(',')]
            rest = rest.replace(mm.group(0), '')
    toks = [t.strip() for t in rest.split(',') if t.strip()]
    dst, srcs = toks[0], toks[1:]
    n = len(srcs)
    sel = mods['op_sel'] or [0] * n
    selh = mods['op_sel_hi'] or [1] * n
    nlo = mods['neg_lo'] or [0] * n
    nhi = mods['neg_hi'] or [0] * n
    dlo, dhi = _halves(dst)
    lo_ops = [('-' if nlo[k] else '') + _halves(srcs[k])[sel[k]] for k in range(n)]
    hi_ops = [('-' if nhi[k] else '') + _halves(srcs[k])[selh[k]] for k in range(n)]
    ins = {'fma': 'v_fma_f32', 'mul': 'v_mul_f32_e64', 'add': 'v_add_f32_e64'}[op]
    return ['\t%s %s, %s\n' % (ins, tmp, ', '.join(lo_ops)), '\t%s %s, %s\n' % (ins, dhi, ', '.join(hi_ops)), '\tv_mov_b32_e32 %s, %s\n' % (dlo, tmp)]


def depack(lines, which='all'):
    """which: all | fma | mul | add | opsel (only instructions carrying op_sel / op_sel_hi) | plain (only those without) | neg (neg_lo / neg_hi)"""
    def wanted(ln):
        m = PK.match(ln)
        if not m:
            return False
        if which in ('fma', 'mul', 'add'):
            return m.group(1) == which
        if which == 'opsel':
            return 'op_sel' in ln
        if which == 'plain':
            return 'op_sel' not in ln and 'neg_' not in ln
        if which == 'neg':
            return 'neg_' in ln
        if which in PATTERNS:
            return m.group(1) == 'fma' and re.search(PATTERNS[which], ln) is not None
        return True
    out, inside, n, in_desc = [], False, 0, False
    for ln in lines:
        if ln.startswith(KERNEL + ':'):
            inside = True
        if inside and wanted(ln):
            out.extend(depack_line(ln, 'v224'))
            n += 1
            continue
        if inside and 's_endpgm' in ln:
            inside = False
        if ln.strip() == '.amdhsa_kernel ' + KERNEL:
            in_desc = True
        if in_desc:
            if '.amdhsa_next_free_vgpr' in ln:
                assert ln.split()[-1] == '224', ln
                ln = ln.replace('224', '228')
            if '.amdhsa_accum_offset' in ln:
                ln = ln.replace('224', '228')
            if '.end_amdhsa_kernel' in ln:
                in_desc = False
        out.append(ln)
    # metadata of the kernel (vgpr_count) follows its .name entry
    for i, ln in enumerate(out):
        if ln.strip() == '.name:           ' + KERNEL:
            for j in range(i, min(i + 12, len(out))):
                if '.vgpr_count:' in out[j]:
                    out[j] = out[j].replace('224', '228')
    return out, n


def transform(lines, name):
    if name.startswith('depack'):
        return depack(lines, name[7:] or 'all')
    out, inside, npk = [], False, 0
    total = 0
    for ln in lines:
        if ln.startswith(KERNEL + ':'):
            inside = True
        if inside and PK.match(ln):
            total += 1
        if inside and 's_endpgm' in ln:
            inside = False
    inside = False
    for ln in lines:
        if ln.startswith(KERNEL + ':'):
            inside = True
        if inside:
            if PK.match(ln):
                npk += 1
                first = npk <= total // 2
                if name == 'nop_before_pk':
                    out.append('\ts_nop 4\n')
                elif name == 'lgkm_before_pk':
                    out.append('\ts_waitcnt lgkmcnt(0)\n')
                elif name == 'vm_before_pk':
                    out.append('\ts_waitcnt vmcnt(0)\n')
                elif (name == 'first_half' and first) or (name == 'second_half' and not first):
                    out.append('\ts_nop 4\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n')
            out.append(ln)
            if name == 'nop_after_ds' and re.match(r'^\s*ds_read', ln):
                out.append('\ts_nop 4\n')
            if 's_endpgm' in ln:
                inside = False
        else:
            out.append(ln)
    return out, total


if __name__ == '__main__':
    names = sys.argv[1:] or ['nop_before_pk', 'lgkm_before_pk', 'vm_before_pk', 'nop_after_ds']
    os.makedirs(WORK, exist_ok=True)
    os.makedirs(os.path.join(ROOT, 'tools', '_ubench'), exist_ok=True)
    base = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++20', '-fPIC', '-fno-gpu-rdc']
    r = run(base + ['-v', '-save-temps', '-c', os.path.join(ROOT, 'dir_amd', 'csrc', 'mano.hip'), '-o', 'mano.o'], cwd=WORK)
    cmds = [ln.strip() for ln in r.stderr.splitlines() if ln.strip().startswith('"')]
    assert len(cmds) == 10, len(cmds)
    run(base + ['-c', os.path.join(ROOT, 'dir_amd', 'csrc', 'core.hip'), '-o', 'core.o'], cwd=WORK)
    asm = os.path.join(WORK, 'mano-hip-amdgcn-amd-amdhsa-gfx950.s')
    orig = open(asm).readlines()
    for name in names:
        mod, total = transform(orig, name)
        open(asm, 'w').writelines(mod)
        for i in (3, 4, 5, 7, 8, 9):          # device assemble, link, bundle; host compile (embeds the new bundle), assemble
            run(shlex.split(cmds[i]), cwd=WORK)
        out = os.path.join(ROOT, 'tools', '_ubench', 'libmano_pk_%s.so' % name)
        run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out, 'mano.o', 'core.o'], cwd=WORK)
        print('%s: %d packed instructions in the kernel, %d lines inserted -> %s' % (name, total, len(mod) - len(orig), out))
    open(asm, 'w').writelines(orig)
