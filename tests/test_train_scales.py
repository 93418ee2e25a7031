"""Host logic of the training convolutions' operand-scale cache (dir_amd/train/conv.py): one cache per owner object, released by end_step,
re-measured every DIR_TRAIN_RECALIBRATE steps -- and a warning when a call site outgrew the headroom its stale scale left.
No GPU, no library call: _site_scale only takes a maximum."""
import warnings

import torch

from dir_amd.train import conv as TC


class Owner:
    pass


def _step(owner, tensors):
    TC.begin_step(owner)
    out = [TC._site_scale(t) for t in tensors]
    TC.end_step()
    return out


def test_scales_are_cached_per_owner_and_released_after_the_step():
    a, b = Owner(), Owner()
    x = torch.full((2, 4, 4, 32), 3.0)
    s1 = _step(a, [x])
    assert _step(a, [x * 8]) == s1                     # second step of the same owner: the cached scale, not a new measurement
    assert _step(b, [x * 8]) == [s1[0] / 8]            # another model measures its own
    assert TC._site_scale(x * 8) == s1[0] / 8          # outside a step nothing is cached
    TC.reset_scales(a)
    assert _step(a, [x * 8]) == [s1[0] / 8]


def test_a_site_that_outgrew_its_scale_is_reported_at_recalibration(monkeypatch):
    monkeypatch.setattr(TC, 'RECALIBRATE', 3)
    o = Owner()
    x = torch.full((1, 2, 2, 32), 1.0)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        _step(o, [x, x])                               # step 1 measures
        _step(o, [x * 1000, x * 2])                    # steps 2, 3 run on the cached scales (site 0 saturates: 1000x > 64x headroom)
        _step(o, [x * 1000, x * 2])
        assert not w
        s = _step(o, [x * 1000, x * 2])                # step 4 = recalibration: site 0 is reported, site 1 (2x) is not
        assert len(w) == 1 and 'call site 0' in str(w[0].message) and issubclass(w[0].category, RuntimeWarning)
        assert s[0] < s[1]
        _step(o, [x * 1000, x * 2])                    # and the new scales are in use without further noise
        assert len(w) == 1
