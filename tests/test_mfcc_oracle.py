"""Pin the MFCC oracle (oracle/mfcc_oracle.py) to the reference's own AudioProcessing.MFCC outputs (G10)."""
import numpy as np
import pytest

from oracle import mfcc_oracle as mo


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_mfcc_oracle_matches_reference(golden, tag):
    g = golden('G10_mfcc')
    sig, rate = g['signal_' + tag], int(g['rate_' + tag])
    out, parts = mo.mfcc(sig, rate, d1=True, d2=True, return_parts=True)
    assert out.shape[0] == int(g['nframes_' + tag])
    n = out.shape[0]
    np.testing.assert_allclose(parts['spec'][[0, 1, n // 2, n - 1]], g['spec_rows_' + tag], rtol=1e-9, atol=1e-7)
    np.testing.assert_allclose(parts['fbank'], g['fbank_' + tag], rtol=1e-9, atol=1e-7)
    np.testing.assert_allclose(parts['energy'], g['energy_' + tag], rtol=1e-10)
    np.testing.assert_allclose(out, g['mfcc39_' + tag], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(mo.mfcc(sig, rate), g['mfcc13_' + tag], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(mo.mfcc(sig, rate, cal_energy=False), g['mfcc13_noenergy_' + tag], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(mo.mfcc(sig, rate, d1=True), g['mfcc26_' + tag], rtol=1e-8, atol=1e-9)


def test_mfcc_quirks_are_really_there(golden):
    """W1 (frame-level scaling) and W3 (saw-tooth filters) are not the textbook forms."""
    assert abs(mo.frame_scale(44)[0] - 0.08) < 1e-12 and abs(mo.frame_scale(44)[-1] - 0.08) < 1e-12
    r = mo.mel_response(16000)
    row = r[10]
    nz = np.nonzero(row)[0]
    assert np.any(np.diff(row[nz[0]:nz[-1] + 1]) < 0)      # drops back to ~0 in the middle: two rising ramps
