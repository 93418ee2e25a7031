"""The shipped kernel-choice tables (dir_amd/tuning/*.json) are well-formed: CPU-side schema check; that a table matches the engine and keeps
the outputs bit-identical is a GPU test (tests/test_gpu_dir.py::test_shipped_throughput_table_applies_and_is_bit_identical)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tuning_tables_are_well_formed():
    from dir_amd.engine import DirEngine
    paths = sorted(glob.glob(os.path.join(ROOT, 'dir_amd', 'tuning', '*.json')))
    assert paths, 'no tuning table shipped'
    for p in paths:
        with open(p) as f:
            t = json.load(f)
        assert isinstance(t['batch'], int) and t['dtype'] in ('bf16', 'f16x3', 'f16') and os.path.basename(p) == 'gfx950_%s_b%d_throughput.json' % (t['dtype'], t['batch'])
        assert len(t['table']) >= 50 and len(t['table']) == len(t['time_tuned_table'])
        for row, row_t in zip(t['table'], t['time_tuned_table']):
            assert len(row) == 6 and row[:5] == row_t[:5] and all(isinstance(v, int) for v in row)
            assert row[5] in DirEngine.CONV_VARIANTS and row_t[5] in DirEngine.CONV_VARIANTS
        m = t['meta']
        assert m['objective'].startswith('time x (socket power - idle power)') and m['idle_w'] > 100
        for r in m['layers']:                       # the measurement the choice was made from travels with the table
            assert r['chosen'] in DirEngine.CONV_VARIANTS and r['us'] > 0 and r['w'] > r['us'] * 0 + 100


def test_power_sampler_degrades_without_rocm_smi():
    """dir_amd/power.py on a box without a GPU / rocm-smi reading: smi_sample() is None (or a well-formed dict where the tool works), the
    sampler thread starts and stops cleanly and returns what it got -- bench.py then reports `power: null` instead of failing."""
    import time
    from dir_amd import power
    v = power.smi_sample()
    assert v is None or (set(v) == {'w', 'sclk', 'cap'} and v['w'] >= 0)
    smp = power.Sampler(skip=0.0, period=0.02).start()
    time.sleep(0.1)
    got = smp.stop()
    assert isinstance(got, list) and (v is not None or got == [])
    assert power.median([], 'w') != power.median([], 'w')          # NaN
    assert power.median([{'w': 1.0}, {'w': 3.0}, {'w': 2.0}], 'w') == 2.0
    e = power.energy_joules()                                         # the amdsmi energy accumulator: None without a GPU, else (joules, seconds)
    assert e is None or (len(e) == 2 and e[0] >= 0)
