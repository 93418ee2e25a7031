"""Energy side of the roofline: runs tools/ubench_energy.hip's loop shapes for a few seconds each while rocm-smi is sampled, and prints watts,
the rate the loop reached, and joules per unit above the spin loop (waves resident, clocks up, nothing switching) and above idle.
usage (GPU box): python tools/ubench_energy.py [seconds = 3]"""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dir_amd import power
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
exe = '/tmp/ubench_energy'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-o', exe, os.path.join(ROOT, 'tools', 'ubench_energy.hip')])
names = {0: 'spin (s_sleep)', 1: 'mfma, operands in registers', 2: 'mfma + 16 ds_read_b128', 3: 'mfma + ds_read + 3 dma', 4: 'ds_read_b128 alone',
         5: 'lds-dma alone (L2 -> LDS)', 6: 'hbm read', 7: 'hbm copy (r + w)', 8: 'read, 96 MB buffer (Infinity Cache)', 9: 'read, 16 MB buffer (L2)'}
rows = []
for mode, data in ((0, 1), (1, 1), (1, 0), (2, 1), (3, 1), (4, 1), (5, 1), (6, 1), (7, 1), (8, 1), (9, 1)):
    smp = power.Sampler(skip=0.6 * SECS, period=0.05).start()
    pr = subprocess.Popen([exe, str(mode), str(SECS + 1.0), str(data)], stdout=subprocess.PIPE, text=True)
    time.sleep(1.2)                                  # process start-up, allocations, clock ramp
    e0 = power.energy_joules()
    time.sleep(max(SECS - 0.6, 0.5))
    e1 = power.energy_joules()
    out = pr.communicate()[0].strip().splitlines()[-1]
    s = smp.stop()
    r = json.loads(out)
    wc = (e1[0] - e0[0]) / (e1[1] - e0[1]) if e0 and e1 else float('nan')        # exact: the socket's energy accumulator over a steady window
    r.update(w=wc if wc == wc else power.median(s, 'w'), w_smi=power.median(s, 'w'), sclk=power.median(s, 'sclk'), name=names[mode] + ('' if data else ' (all-zero data)'))
    rows.append(r)
spin = rows[0]['w']
print('idle %.0f W (dir_amd.power.IDLE_W); spin loop %.0f W at %.0f MHz  (watts: energy accumulator over a steady window when amdsmi has it; rocm-smi in brackets)' % (power.IDLE_W, spin, rows[0]['sclk']))
for r in rows[1:]:
    pj_spin = (r['w'] - spin) / r['rate'] * 1e12
    pj_idle = (r['w'] - power.IDLE_W) / r['rate'] * 1e12
    scale = {'FLOP': (1e12, 'TFLOP/s'), 'LDS bytes': (1e12, 'TB/s'), 'DMA bytes': (1e12, 'TB/s'), 'HBM bytes': (1e12, 'TB/s')}[r['unit']]
    print('%-34s %6.0f W [%4.0f] %5.0f MHz  %8.2f %-8s  %6.2f pJ/%s above spin, %6.2f above idle' % (
        r['name'], r['w'], r['w_smi'], r['sclk'], r['rate'] / scale[0], scale[1], pj_spin, r['unit'].split()[0] if r['unit'] != 'FLOP' else 'FLOP', pj_idle))
