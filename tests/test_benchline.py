"""bench.py's last stdout line must stay small enough for the driver to parse (round 3: 20 KB -> `parsed: null`).  Host logic only."""
import json
import os

from dir_amd import benchline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _canned():
    with open(os.path.join(ROOT, 'profiles', 'r03_f_bench_line.txt')) as f:
        return json.load(f)


def test_compact_line_is_small_and_complete():
    detail = _canned()
    assert len(json.dumps(detail)) > 15000                      # the record that went unparsed
    line = benchline.compact(detail)
    s = json.dumps(line)
    assert len(s) <= benchline.LIMIT < 6000
    assert json.loads(s) == line                                  # strict JSON, one line
    assert '\n' not in s
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data'):
        assert line[k] == detail[k]
    assert line['config']['workload'] and line['config']['world_size_observed'] == 1
    r = line['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert r[k] == detail['roofline'][k]
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert set(r['by_class']) == {'hbm', 'mfma'}
    c = line['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind'):
        assert c[k] == detail['cpu_baseline'][k]
    for k in ('fp32_mode', 'parity_mode_f16x3', 'fp16_mode'):
        assert line[k]['images_per_sec'] == detail[k]['images_per_sec']
    assert line['train_step']['seconds_per_step'] == detail['train_step']['seconds_per_step']


def test_compact_line_survives_growth():
    """whatever later rounds add to the record, the line stays under the limit"""
    detail = _canned()
    detail['roofline']['kernels'] = detail['roofline']['kernels'] * 4
    detail['config']['workload'] = 'x' * 5000
    detail['roofline']['kernel'] = 'k' * 5000
    detail['new_mode'] = {'note': 'n' * 10000}
    assert len(json.dumps(benchline.compact(detail))) <= benchline.LIMIT


def test_compact_line_without_optional_parts():
    detail = {k: v for k, v in _canned().items() if k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'config')}
    line = benchline.compact(detail)
    assert line['roofline'] is None and line['cpu_baseline'] is None
