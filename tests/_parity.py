"""Parity bookkeeping for the GPU tests: every comparison against the oracle that the north star bounds (1e-4 relative on
log-likelihoods and gamma / xi occupancies in f32, and what follows from them: the E-step statistics and the re-estimated
model) goes through `hold`, which ASSERTS the bound and RECORDS the measured worst case.  The session writes the records to
profiles/r04_parity_report.json (and gpurun_out/, which is what travels back from the GPU box), so that the numbers the
asserts saw are on file, not only printed.

A record: config / quantity -> {bound: {rtol, atol}, max_abs, max_rel (over entries with |want| > atol / rtol, i.e. where the
relative term of the bound is the binding one), used (max over entries of |got - want| / (atol + rtol |want|): <= 1 passes),
n}.  Repeated calls with the same key keep the worst of each figure."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = {}


def measure(got, want, rtol, atol=0.0):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin), 'finite pattern differs'
    g, w = got[fin], want[fin]
    if g.size == 0:
        return dict(max_abs=0.0, max_rel=0.0, used=0.0, n=0)
    err = np.abs(g - w)
    tol = atol + rtol * np.abs(w)
    used = float((err / np.where(tol > 0, tol, 1.0))[tol > 0].max()) if (tol > 0).any() else 0.0
    if (tol == 0).any() and err[tol == 0].max() > 0:
        used = float('inf')
    big = np.abs(w) > (atol / rtol if rtol > 0 else np.inf)
    return dict(max_abs=float(err.max()), max_rel=float((err[big] / np.abs(w[big])).max()) if big.any() else 0.0, used=used, n=int(g.size))


def hold(config, quantity, got, want, rtol, atol=0.0, note=None):
    """assert |got - want| <= atol + rtol |want| on every finite entry (same finite pattern) and record the measured worst case."""
    m = measure(got, want, rtol, atol)
    rec = REPORT.setdefault(config, {}).setdefault(quantity, dict(bound=dict(rtol=rtol, atol=atol), max_abs=0.0, max_rel=0.0, used=0.0, n=0))
    rec['bound'] = dict(rtol=max(rec['bound']['rtol'], rtol), atol=max(rec['bound']['atol'], atol))
    for k in ('max_abs', 'max_rel', 'used'):
        rec[k] = max(rec[k], m[k])
    rec['n'] += m['n']
    if note:
        rec['note'] = note
    msg = '%s / %s: %.3g of the bound (rtol %g, atol %g): max |d| %.3e, max relative %.3e' % (
        config, quantity, m['used'], rtol, atol, m['max_abs'], m['max_rel'])
    if os.environ.get('POCCALA_PARITY_SOFT'):          # discovery runs: record every violation instead of stopping at the first
        if m['used'] > 1.0:
            rec['violated'] = True
            print('PARITY VIOLATION ' + msg)
        return m
    assert m['used'] <= 1.0, msg
    return m


def note(config, key, value):
    """a measured figure that is not a bounded comparison (a flip rate, a count)."""
    REPORT.setdefault(config, {})[key] = value


def write():
    if not REPORT:
        return
    body = dict(what='measured worst cases of the GPU parity tests (tests/_parity.py): every entry was asserted against its bound in the run that wrote this file',
                contract='BASELINE.json north_star: log-likelihoods and gamma / xi occupancies within 1e-4 relative in f32, Viterbi bit-exact',
                configs=REPORT)
    for d in (os.path.join(ROOT, 'profiles'), os.path.join(ROOT, 'gpurun_out')):
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, 'r04_parity_report.json')
            old = {}
            if os.path.exists(path) and os.environ.get('POCCALA_PARITY_MERGE'):
                old = json.load(open(path)).get('configs', {})
            merged = dict(old)
            merged.update(REPORT)
            body['configs'] = merged
            with open(path, 'w') as f:
                json.dump(body, f, indent=1, sort_keys=True)
        except OSError:
            pass
