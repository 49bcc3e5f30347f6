"""A13 at the BASELINE mixture counts: the E-step statistics of Clustering.GMM.update_acc (StatisticalModel/Clustering.py:653-680)
per mixture against the oracle at M = 256 (C2) and M = 2048 (the C4 shard, the bench workload).

The small-M end-to-end tests (test_gpu_parity.py::test_estep_*) cannot reach what only exists at real M: the eight
256-mixture slice workgroups of a state, several state groups alternating between the two tile-image buffers, last-tile
images of long active lists.  Here the device's OWN ln gamma_t(j) and ln b_j(o_t) go into the oracle's update_acc for every
occurrence of a handful of states, so the comparison isolates the accumulate kernels: acc, alpha_acc, mean_acc, cov_acc per
mixture, f32-class default (rtol 1e-4 -- the north-star bound --, atol 1e-6 of the state's largest entry) and PCL_F64 (1e-9);
the measured worst cases go to gpurun_out/r05_parity_report.json (tests/_parity.py; profiles/ on request)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

S = 5


def pick_states(alpha_acc, rng, extra=1):
    """the least and the most occupied state, the first and the last state id seen (first / last state group), and
    `extra` random ones."""
    seen = np.flatnonzero(alpha_acc > 0)
    occ = alpha_acc[seen]
    picks = [int(seen[np.argmin(occ)]), int(seen[np.argmax(occ)]), int(seen[0]), int(seen[-1])]
    picks += [int(j) for j in rng.choice(seen, size=extra, replace=False)]
    out = []
    for j in picks:
        if j not in out:
            out.append(j)
    return out


def run_case(cfg_name, prec, image_mb, peaked, n_extra=1, min_groups=0):
    from poccala_amd import Engine, PCL_F32, PCL_F64, synth
    from _oracle_pool import state_statistics
    from _parity import cov_acc_atol, hold
    P = PCL_F32 if prec == 'f32' else PCL_F64
    c = synth.CONFIGS[cfg_name]
    from _models import full_size_model
    mean, var, w, trans = full_size_model(c, 1) if cfg_name == 'C4shard' else synth.make_model(c['units'], c['M'], c['D'], seed=1)
    labels = synth.make_labels(c['U'], c['L'], c['units'], seed=2)
    if peaked:
        frames = synth.make_peaked_frames(labels, c['T'], mean, var, seed=5)
        lens = np.full(c['U'], c['T'], dtype=np.int32)
        begin = np.arange(c['U'], dtype=np.int64) * c['T']
    else:
        frames, lens, begin = synth.make_frames(c['U'], c['T'], c['D'], seed=0)
    old = os.environ.get('PCL_ACC_IMAGE_MB')
    if image_mb is not None:
        os.environ['PCL_ACC_IMAGE_MB'] = str(image_mb)           # read per accumulate call: forces many state groups
    eng = Engine(0)
    try:
        eng.enable_timing(True)
        eng.load_model(mean, var, w)
        eng.load_frames(frames)
        eng.load_units(np.stack(trans))
        b = eng.label_batch(labels, lens, begin)
        b.score(P)
        b.forward_backward(fix_pi=False)
        eng.stats_zero()
        eng.kernel_time('acc_consume')
        b.accumulate(P)
        st = eng.stats_download()
        groups = eng.kernel_time('acc_consume')[1]
        lg, B = b.get('lgamma'), b.get('B')
        b.close()
    finally:
        eng.close()
        if image_mb is not None:
            if old is None:
                del os.environ['PCL_ACC_IMAGE_MB']
            else:
                os.environ['PCL_ACC_IMAGE_MB'] = old
    if prec == 'f32':
        assert groups >= min_groups, 'expected the state groups to alternate between the image buffers, got %d' % groups
    picks = pick_states(st['alpha_acc'], np.random.default_rng(11), n_extra)
    jobs, owner = [], []
    e = S - 2
    for j in picks:
        unit, k = divmod(j, e)
        for u, lab in enumerate(labels):
            for pos in np.flatnonzero(np.asarray(lab) == unit):
                row = 1 + int(pos) * e + k
                x = frames[begin[u]:begin[u] + lens[u]].astype(np.float64)
                jobs.append((x, lg[u][row].copy(), B[u][row].copy(), mean[j], var[j], w[j]))
                owner.append(j)
    res = state_statistics(jobs)
    rt, at = (1e-4, 1e-6) if prec == 'f32' else (1e-9, 1e-13)
    worst = {}
    tag = '%s accumulate %s%s' % (cfg_name, prec, ' peaked' if peaked else '')
    for j in picks:
        ref = dict(acc=0.0, alpha_acc=0.0, mean_acc=0.0, cov_acc=0.0)
        for o, r in zip(owner, res):
            if o == j:
                for key in ref:
                    ref[key] = ref[key] + r[key]
        for key in ('acc', 'alpha_acc', 'mean_acc', 'cov_acc'):
            got, want = np.asarray(st[key][j]), np.asarray(ref[key])
            scale = float(np.abs(want).max())
            bound = at * scale
            if key == 'cov_acc' and prec == 'f32':               # + the raw-moment term (tests/_parity.py:cov_acc_atol)
                bound = cov_acc_atol(np.asarray(ref['acc']), mean[j], var[j], bound)
            hold(tag, key, got, want, rt, bound)
            big = np.abs(want) > 1e-3 * scale
            if big.any():
                worst[key] = max(worst.get(key, 0.0), float((np.abs(got - want)[big] / np.abs(want)[big]).max()))
    print('%s %s%s: %d states (%s), %d occurrences, %d state groups; worst relative error on entries > 1e-3 of the state max: %s'
          % (cfg_name, prec, ' peaked' if peaked else '', len(picks), picks, len(jobs), groups,
             ', '.join('%s %.1e' % kv for kv in worst.items())))


@pytest.mark.parametrize('prec,image_mb,peaked', [('f32', 64, False), ('f32', None, True), ('f64', None, False)])
def test_statistics_per_mixture_c2(prec, image_mb, peaked):
    """M = 256: one slice workgroup per state; PCL_ACC_IMAGE_MB = 64 splits the 700 MB of tile images into >= 3 state
    groups, so both image buffers are reused while the producer of the next group runs beside the consumer."""
    run_case('C2', prec, image_mb, peaked, n_extra=2, min_groups=3 if image_mb else 1)


@pytest.mark.parametrize('prec,peaked', [('f32', False)] + ([('f32', True), ('f64', False)] if os.environ.get('POCCALA_SOAK') else []))      # (peaked posteriors and f64 at M = 2048: the soak; the suite holds both at C2)
def test_statistics_per_mixture_c4_shard(prec, peaked):
    """M = 2048, J = 3000: the bench workload (8 slice workgroups per state, 7 state groups at the default 2 GB image budget)."""
    run_case('C4shard', prec, None, peaked, n_extra=1, min_groups=0 if peaked else 3)


def test_compaction_by_rows_keeps_the_bits():
    """The active-frame lists built by rows (a wave owns 8 rows of an utterance) are the lists the segment kernels built, entry for
    entry: statistics, the re-estimated model, its conditioning and the next scores of a ragged problem with duplicate label rows and
    split states hash to the same value under PCL_ACC_ROWS=0 and =1 (the knob is read once per process: two processes)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for rows in ('0', '1'):
        env = dict(os.environ, PCL_ACC_ROWS=rows)
        r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'acc_hash.py')], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith(('pass ', 'model '))]
        assert len(lines) == 4, r.stdout
        outs.append(lines)
    assert outs[0] == outs[1]
