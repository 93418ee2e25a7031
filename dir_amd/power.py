"""Socket power / shader clock of the visible GPU from `rocm-smi --json`, and the socket's energy accumulator from the amdsmi Python binding
(15.3 uJ counts: exact joules over a window, no averaging lag) -- tuning and reporting aids (DirEngine.autotune_energy, bench.py's `power`
object).  Nothing on the forward path imports this module."""
import json
import shutil
import statistics
import subprocess
import threading
import time

IDLE_W = 243.0       # package power of an idle MI355X as rocm-smi reports it on the boxes this was measured on; measured live when possible


def smi_sample():
    """{'w': socket power (W), 'sclk': shader clock (MHz), 'cap': power cap (W)} or None (no rocm-smi, no permission, unparsable output)"""
    exe = shutil.which('rocm-smi') or '/opt/rocm/bin/rocm-smi'
    try:
        r = subprocess.run([exe, '--showpower', '--showclocks', '--showmaxpower', '--json'], capture_output=True, text=True, timeout=10)
        d = next(iter(json.loads(r.stdout.strip().splitlines()[-1]).values()))
        num = lambda v: float(''.join(c for c in str(v) if c.isdigit() or c == '.'))                      # noqa: E731
        return {'w': num(next(v for k, v in d.items() if 'Package Power (W)' in k and 'Max' not in k)),
                'sclk': num(d.get('sclk clock speed:', '0')), 'cap': num(d.get('Max Graphics Package Power (W)', '0'))}
    except Exception:
        return None


class Sampler(object):
    """Samples smi_sample() from a thread between start() and stop(); samples taken before `skip` seconds are dropped (the reading is a
    moving average).  stop() returns the list of samples."""

    def __init__(self, skip, period=0.05):
        self.skip, self.period, self.samples = skip, period, []
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        t0 = time.perf_counter()
        while not self._stop.is_set():
            v = smi_sample()
            if v is not None and time.perf_counter() - t0 > self.skip:
                self.samples.append(v)
            self._stop.wait(self.period)

    def start(self):
        self._th.start()
        return self

    def stop(self):
        self._stop.set()
        self._th.join()
        return self.samples


def median(samples, key):
    return statistics.median(v[key] for v in samples) if samples else float('nan')


_smi = None          # (module, handle) once initialised, False when unavailable


def energy_joules():
    """(joules accumulated by the socket's energy counter, its timestamp in seconds) from amdsmi, or None when the binding / the counter is
    not available.  Differences of two readings are exact energies (counter resolution 15.3 uJ); the first visible GPU is read."""
    global _smi
    if _smi is None:
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            _smi = (amdsmi, hs[0]) if hs else False
        except Exception:
            _smi = False
    if not _smi:
        return None
    try:
        d = _smi[0].amdsmi_get_energy_count(_smi[1])
        acc = d.get('energy_accumulator', d.get('power'))
        return float(acc) * float(d['counter_resolution']) * 1e-6, float(d['timestamp']) * 1e-9
    except Exception:
        return None
