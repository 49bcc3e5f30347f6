"""CPU-side checks of bench.py's record (VERDICT r4 next #1d): the `roofline` object's fractions follow from its own numbers, the
committed PMC traffic figure is served exactly when the profiled kernel is the current kernel, the CPU baseline legs the line
carries exist in the oracle and agree with it, and the newest committed bench line has the schema the contract prescribes."""
import glob
import json
import os
import shutil

import numpy as np
import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_roofline_fractions_follow_from_their_numbers():
    pairs, M, D, ms = 18273600, 2048, 39, 14.4
    r = bench.make_roofline(7, ms, 20, pairs, 18432000, M, D, 2.136e9, 5.4e9, dict(FETCH_SIZE_bytes=2.4e9, WRITE_SIZE_bytes=0.6e9), 0.8, 0.47)
    flop = pairs * M * (3 * D + 4)
    assert r['flop_per_launch'] == flop and r['unit'] == 'TFLOP/s' and r['bound'] == 'mfma'
    assert r['achieved'] == pytest.approx(flop / (ms * 1e-3) / 1e12)
    assert r['peak'] == pytest.approx(2516.6 / 3) and r['frac'] == pytest.approx(r['achieved'] / r['peak'])
    assert r['frac_of_f16_dense_peak'] == pytest.approx(r['achieved'] / 2516.6)
    assert r['executed_mfma_tflops'] == pytest.approx(pairs * M * 480 / (ms * 1e-3) / 1e12)
    assert r['frac_executed'] == pytest.approx(r['executed_mfma_tflops'] / 2516.6)
    assert r['frac_of_f16_dense_peak'] < r['frac'] < 1 and r['frac_executed'] == pytest.approx(r['frac_of_f16_dense_peak'] * 480 / 121)
    assert r['traffic'] == 5.4e9 and r['traffic_over_algorithmic'] == pytest.approx(5.4 / 2.136)
    assert 'denominator' in ''.join(r) and '2516.6' in r['frac_denominator']
    for k in ('kernel', 'kernel_avg_ms', 'launches', 'scored_pairs', 'label_pairs', 'hbm_frac', 'fb_kernel_avg_ms', 'fb_kernel_alone_ms', 'note'):
        assert k in r
    # no launches measured: nothing is made up
    r0 = bench.make_roofline(7, None, 0, pairs, pairs, M, D, 2.136e9, None, None, 0.0, None)
    assert r0['achieved'] is None and r0['frac'] is None and r0['frac_executed'] is None and r0['traffic'] is None


def _tree(tmp_path, summary_sha):
    (tmp_path / 'poccala_amd' / 'csrc').mkdir(parents=True)
    (tmp_path / 'profiles').mkdir()
    for f in ('gmm_score_split.hip', 'Makefile'):
        shutil.copy(os.path.join(ROOT, 'poccala_amd', 'csrc', f), tmp_path / 'poccala_amd' / 'csrc' / f)
    (tmp_path / 'profiles' / 'r90_bench_summary.txt').write_text(
        '# older round\nkernel_code_sha16 gmm_score_split16_kernel 0000000000000000\n'
        'gmm_score_split16_kernel<39,2>   FETCH_SIZE   n=24 per-dispatch=1000\ngmm_score_split16_kernel<39,2>   WRITE_SIZE   n=24 per-dispatch=100\n')
    (tmp_path / 'profiles' / 'r91_bench_summary.txt').write_text(
        '# newest round\nkernel_code_sha16 gmm_score_split16_kernel %s\n'
        'gmm_score_split16_kernel<39,2>   FETCH_SIZE   n=24 per-dispatch=2000000\ngmm_score_split16_kernel<39,2>   WRITE_SIZE   n=24 per-dispatch=500000\n' % summary_sha)


def test_committed_traffic_is_served_for_the_current_kernel_and_withheld_otherwise(tmp_path, monkeypatch):
    sha = bench.scoring_kernel_sha16()
    _tree(tmp_path, sha)
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    assert bench.scoring_kernel_sha16() == sha                      # (same kernel text and flags in the copy)
    t, raw = bench.committed_traffic()
    assert t == pytest.approx(2 * 2000000 * 1024.0 + 500000 * 1024.0)           # the NEWEST summary; 2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes
    assert raw['file'] == 'profiles/r91_bench_summary.txt' and raw['kernel_code_sha16'] == sha
    # the kernel body changes: the figure is withheld and the record says why
    src = tmp_path / 'poccala_amd' / 'csrc' / 'gmm_score_split.hip'
    text = src.read_text()
    src.write_text(text.replace('namespace {', 'namespace {\nconstexpr int A_NEW_CONSTANT = 1;', 1))
    t, raw = bench.committed_traffic()
    assert t is None and 'stale' in raw
    # a comment, white space, or the launch side of the file: the kernel is the same kernel
    src.write_text(text.replace('namespace {', 'namespace {\n// a remark\n\n', 1) + '\n// launch-side edit\nstatic int unused_launch_helper() { return 0; }\n')
    t, raw = bench.committed_traffic()
    assert t is not None and raw['kernel_code_sha16'] == sha


def test_cpu_baseline_legs_are_the_oracle():
    from oracle import poccala_oracle as po
    rng = np.random.default_rng(3)
    M, D, T = 300, 39, 37
    mean, var = rng.standard_normal((M, D)), rng.uniform(0.5, 2.0, (M, D))
    w = rng.dirichlet(np.ones(M))
    w[5] = 0.0
    x = rng.standard_normal((T, D))
    with np.errstate(divide='ignore'):
        ref = po.gmm_point(x, mean, var, w)
        assert np.array_equal(po.gmm_point_blocked(x, mean, var, w), ref)                 # the same arithmetic, blocked: the same bits
        assert np.array_equal(po.gmm_point_blocked(x, mean, var, w, m_chunk=7, t_chunk=5), ref)
        np.testing.assert_allclose(po.gmm_point_gemm(x, mean, var, w), ref, rtol=1e-11)   # the expanded form: another order of operations


def _validate_line(d):
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
              'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['metric'] == 'frames/sec GMM-score+fwd-bwd, 39-d MFCC, 2048-mix' and d['unit'] == 'frames/s' and d['vs_baseline'] is None
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    assert d['value'] == pytest.approx(d['config']['frames_per_step_total'] / (d['ms_per_step'] * 1e-3), rel=1e-6)
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'kernel_avg_ms'):
        assert k in r, k
    assert r['frac'] == pytest.approx(r['achieved'] / r['peak'], rel=1e-9)
    assert r['kernel_avg_ms'] <= d['ms_per_step'] * 1.02            # the dominant kernel fits inside the step it is part of
    c = d['cpu_baseline']
    if d['n_gpus'] == 1 and c is not None:
        for k in ('value', 'unit', 'cores', 'kind', 'sample', 'gemm_value', 'faithful_value'):
            assert k in c, k
        assert c['kind'] in ('port', 'reference') and c['cores'] >= 1


def test_newest_committed_bench_line_has_the_contract_schema():
    lines = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_line.json')))
    lines = [f for f in lines if 'f32mfma' not in f and 'valu' not in f]
    assert lines
    d = json.load(open(lines[-1]))
    _validate_line(d)
    rnd = int(os.path.basename(lines[-1])[1:3])
    if rnd >= 5:                                                     # round 5 on: the whole record is inside `roofline`
        r = d['roofline']
        for k in ('frac_of_f16_dense_peak', 'frac_executed', 'frac_denominator', 'sustained_value', 'strict_f32_value', 'strict_f32_frac',
                  'pcie_inclusive_value', 'fresh_batches_value'):
            assert k in r, k
        assert r['traffic'] is not None and r['traffic_raw']['kernel_code_sha16'], 'the line was printed without a PMC traffic figure'
        assert r['frac_of_f16_dense_peak'] == pytest.approx(r['achieved'] / 2516.6, rel=1e-9)
        c = d['cpu_baseline']
        assert c['value'] == max(c['gemm_value'], c['vectorised_value']) and c['value_leg'] in ('gemm', 'vectorised')
        # the line's traffic figure comes from the summary committed beside it, taken from the kernel the line ran
        summ = open(os.path.join(ROOT, 'profiles', 'r%02d_bench_summary.txt' % rnd)).read()
        assert r['traffic_raw']['kernel_code_sha16'] in summ
