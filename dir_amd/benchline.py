"""The ONE JSON line bench.py prints last, cut down from the full measurement record.

Round 3's line had grown to 20 KB (three per-kernel tables, the power object, four sub-mode records) and no longer fitted the window the
driver reads bench.py's stdout through: the record was unparsed.  bench.py now writes the full record to `bench_detail.json` (and to
stderr) and prints `compact(detail)` -- at most LIMIT (5000) characters, every contract key of SURVEY.md section 8(d) / BASELINE.md section 4
plus `roofline` and `cpu_baseline` and one-number summaries of the sub-modes -- as the LAST stdout line.

Pure host logic (no torch, no GPU): tests/test_benchline.py holds it to round 3's kept 20 KB record.
"""
import json

LIMIT = 5000        # characters of json.dumps(compact(detail)); the driver parsed 6.9 KB in round 2 and lost 20 KB in round 3 (its window: 8000)

_TOP = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')
_CONFIG = ('workload', 'batch_per_gpu', 'graph', 'forwards_in_flight', 'ms_per_forward_one_in_flight', 'conv_tuning', 'tunings_bit_identical',
           'outputs_finite', 'overlapped_equals_one_at_a_time', 'world_size_observed', 'backend', 'timed_regions', 'region_ms_per_step', 'region_spread',
           'statistic', 'weights_kind', 'parity', 'images_per_sec_by_mode', 'parity_by_mode')
_ROOF = ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source', 'frac_mfma', 'frac_hbm', 'alg_bytes_per_launch',
         'alg_gflop_per_launch', 'launches_per_step', 'avg_launch_us', 'all_conv_ms_per_step', 'all_kernels_ms_per_step', 'library_calls_per_step',
         'eager_bracket_overhead_us_per_call')


def _pick(d, keys):
    return {k: d[k] for k in keys if k in d}


def _short(s, n):
    if isinstance(s, str) and len(s) > n:
        return s[:n - 3] + '...'
    return s


def _by_class(bc):
    if not bc:
        return bc
    return {k: _pick(v, ('launches_per_step', 'ms_per_step', 'achieved', 'unit', 'frac')) for k, v in bc.items()}


def _mode(rec):
    """a sub-mode record -> images/sec + ms only (+ its own roofline fraction when it has one)"""
    if not rec:
        return None
    out = _pick(rec, ('images_per_sec', 'ms_per_step', 'forwards_in_flight', 'ms_per_forward_one_in_flight', 'batch_per_gpu', 'seconds_per_step',
                      'launches_per_step', 'frac_of_f16x3_mfma_peak', 'pad_waste', 'f16_storage_images_per_sec', 'f16_storage_ms_per_step', 'fp16_images_per_sec', 'fp16_ms_per_step',
                      'headline_dtype_alternating_ms_per_step', 'graph_captured_seconds_per_step', 'eager_seconds_per_step'))
    r = rec.get('roofline')
    if r:
        out['roofline_frac'] = r.get('frac')
        out['roofline_bound'] = r.get('bound')
    return out


def compact(detail, detail_path='bench_detail.json'):
    line = _pick(detail, _TOP)
    cfg = detail.get('config') or {}
    c = _pick(cfg, _CONFIG)
    c['workload'] = _short(c.get('workload'), 170)
    c['conv_tuning'] = _short(c.get('conv_tuning'), 90)
    if c.get('parity'):
        c['parity'] = dict(c['parity'], vs=_short(c['parity'].get('vs'), 60))
    line['config'] = c
    if detail.get('parity'):                                 # the headline mode against the reference golden, rows of the timed batch
        line['parity'] = dict(detail['parity'], vs=_short(detail['parity'].get('vs'), 60))
    roof = detail.get('roofline')
    if roof:
        r = _pick(roof, _ROOF)
        r['kernel'] = _short(r.get('kernel'), 190)
        r['traffic_source'] = _short(r.get('traffic_source'), 120)
        r['by_class'] = _by_class(roof.get('by_class'))
        tt = roof.get('time_tuned_table')
        if tt:
            r['time_tuned_table'] = dict(_pick(tt, ('frac_mfma', 'frac_hbm', 'avg_launch_us', 'all_conv_ms_per_step')), by_class=_by_class(tt.get('by_class')))
        ov = roof.get('overlapped')
        if ov:
            r['overlapped'] = _pick(ov, ('achieved', 'unit', 'frac'))
            fif = ov.get('four_in_flight_profile')
            if fif:                                          # per-launch mean under contention, from the tracked four-in-flight kernel trace
                r['overlapped']['four_in_flight_profile'] = _pick(fif, ('source', 'avg_launch_us', 'launches_per_forward', 'conv_ms_per_forward'))
        mc = roof.get('measured_ceilings')
        if mc:
            r['measured_ceilings'] = {k: v for k, v in mc.items() if k != 'source'}
        tk = roof.get('token_path')
        if tk:
            r['token_path'] = tk
        ks = roof.get('kernels') or []
        # the five kernels that cost most per step, one short row each (the whole table is in the detail file)
        r['top_kernels'] = [[k['kernel'][:28], k['calls_per_step'], k['avg_us'], k['bound'], k['frac']] for k in ks[:5]]
        r['top_kernels_columns'] = 'kernel, calls_per_step, avg_us, bound, frac'
        line['roofline'] = r
    else:
        line['roofline'] = None
    cpu = detail.get('cpu_baseline')
    if cpu:
        cb = _pick(cpu, ('value', 'unit', 'cores', 'kind', 'os_cpu_count', 'cpu_model'))
        cb['sample'] = _short(cpu.get('sample'), 120)
        to = cpu.get('torch_ops')
        if to:
            cb['torch_ops'] = {k: (v.get('images_per_sec') if isinstance(v, dict) else v) for k, v in to.items() if k != 'note'}
        line['cpu_baseline'] = cb
    else:
        line['cpu_baseline'] = None
    pw = detail.get('power')
    if pw:
        p = _pick(pw, ('socket_w', 'cap_w', 'sclk_mhz', 'joules_per_step'))
        ec = pw.get('energy_counter')
        if ec:
            p['joules_per_step_energy_counter'] = ec.get('joules_per_step')
        line['power'] = p
    for k in ('fp32_mode', 'parity_mode_f16x3', 'fp16_mode', 'fp16_storage_mode', 'bf16_storage_mode', 'train_step', 'without_proj_feat', 'config5_hrnet'):
        if detail.get(k):
            line[k] = _mode(detail[k])
    pg = detail.get('pgcn')
    if pg:
        line['pgcn'] = {k: ({kk: v[kk] for kk in ('us_per_stack_pair', 'alg_gbps', 'frac_of_hbm_peak')} if isinstance(v, dict) else None) for k, v in pg.items() if k.startswith('B=')}
        line['pgcn']['target'] = 0.6
    line['detail'] = detail_path
    # never over the limit: drop the optional parts, least important first
    for drop in (('roofline', 'top_kernels'), ('roofline', 'top_kernels_columns'), ('roofline', 'time_tuned_table'), ('power',), ('without_proj_feat',),
                 ('roofline', 'traffic_source'), ('fp16_mode',), ('cpu_baseline', 'sample'), ('roofline', 'kernel'), ('config5_hrnet',), ('pgcn',),
                 ('config', 'region_ms_per_step')):
        if len(json.dumps(line)) <= LIMIT:
            break
        o = line
        for k in drop[:-1]:
            o = o.get(k) or {}
        o.pop(drop[-1], None)
    return line
