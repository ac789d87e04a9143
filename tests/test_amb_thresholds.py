"""The erf-free ambiguous-order test (thresholds found by bisection on the host
libm) against the oracle's restatement of ref algorithms.c:175-193."""
import ctypes as C

import numpy as np
import pytest

from helpers import hostsim
from oracle.oracle_py import lib


@pytest.mark.parametrize("pcutoff", [0.01, 0.0, 0.2, 0.3, 0.49999, 0.5, 0.7, -0.5, 1e-6, 1e-30])
def test_threshold_form_equals_erf_pipeline(pcutoff):
    H, L = hostsim(), lib()
    tp, tn = C.c_float(), C.c_float()
    H.hs_amb_thresholds(pcutoff, C.byref(tp), C.byref(tn))
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.normal(0, 3, 4000), rng.normal(0, 0.2, 2000),
                         [0.0, -0.0, np.inf, -np.inf, np.nan, 1e30, -1e30]]).astype(np.float32)
    for t in (tp.value, tn.value):      # probe around the step
        if t >= 0 and np.isfinite(t):
            b = np.float32(t)
            xs = np.concatenate([xs, [b, np.nextafter(b, np.float32(np.inf)), -b,
                                      -np.nextafter(b, np.float32(np.inf)),
                                      np.nextafter(b, np.float32(0)), -np.nextafter(b, np.float32(0))]])
    for x in xs.astype(np.float32):
        want = bool(L.ora_ambiguous_from_interval(float(x), pcutoff))
        got = (x <= tp.value) if x >= 0 else (-x <= tn.value)
        assert bool(got) == want, (float(x), tp.value, tn.value)


def test_pairs_against_oracle():
    H, L = hostsim(), lib()
    rng = np.random.default_rng(2)
    n = 30000
    d1 = rng.integers(-5000, 5000, n); d2 = d1 + rng.integers(-60, 60, n)
    big = rng.random(n) < 0.1
    d1 = np.where(big, d1 * (1 << 40) + rng.integers(0, 1 << 30, n), d1)
    s1 = (rng.random(n) * 30).astype(np.float32); s2 = (rng.random(n) * 30).astype(np.float32)
    s1[rng.random(n) < 0.02] = 0; s2[rng.random(n) < 0.02] = 0
    for pc in (0.01, 0.0, 0.25):
        tp, tn = C.c_float(), C.c_float()
        H.hs_amb_thresholds(pc, C.byref(tp), C.byref(tn))
        for i in range(n):
            a = H.hs_ambiguous(int(d1[i]), float(s1[i]), int(d2[i]), float(s2[i]), tp.value, tn.value)
            b = L.ora_ambiguousorder(int(d1[i]), float(s1[i]), int(d2[i]), float(s2[i]), pc)
            assert bool(a) == bool(b), (i, d1[i], s1[i], d2[i], s2[i])
