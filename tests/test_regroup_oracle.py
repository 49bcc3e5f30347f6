"""The oracle's restatement of the post-alignment regrouping against golden G12, which was made by
running the reference's own discriminate / __eq_segment / __get_gmmdata (tests/golden/make_golden_regroup.py)."""
import os

import numpy as np

from oracle import poccala_oracle as po

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'G12_regroup.npz'))


def test_eq_segment_g_matches_reference():
    for n in G['g_lens']:
        data = G['g_data_%d' % n]
        sl = po.eq_segment_g(data, 3)
        assert [len(s) for s in sl] == list(G['g_sizes_%d' % n])
        for k in range(3):
            np.testing.assert_array_equal(np.asarray(sl[k]).reshape(-1, 3), G['g_slice_%d_%d' % (n, k)])


def test_eq_segment_e_matches_reference():
    blocks = po.eq_segment_e(G['e_data'], list(G['e_label']))
    assert [u for u, _ in blocks] == list(G['e_units'])
    for i, (_, d) in enumerate(blocks):
        np.testing.assert_array_equal(d, G['e_block_%d' % i])


def test_discriminate_and_get_gmmdata_match_reference():
    for ci in range(int(G['n_cases'])):
        seq, data = G['d_seq_%d' % ci], G['d_data_%d' % ci]
        k_of_t = po.regroup_frame_states(seq, 3)
        for unit in sorted(set(seq)):
            runs = po.discriminate(unit, seq)
            assert len(runs) == int(G['d_nruns_%d_%s' % (ci, unit)])
            for ri, loc in enumerate(runs):
                np.testing.assert_array_equal(loc, G['d_loc_%d_%s_%d' % (ci, unit, ri)])
            g = po.get_gmmdata([data[loc] for loc in runs], 3)
            for k in range(3):
                ref = G['d_gmm_%d_%s_%d' % (ci, unit, k)]
                np.testing.assert_array_equal(np.asarray(g[k]).reshape(-1, 3), ref)
                # the per-frame form selects exactly the same frames, in time order
                sel = data[(seq == unit) & (k_of_t == k)]
                np.testing.assert_array_equal(sel.reshape(-1, 3), ref)
