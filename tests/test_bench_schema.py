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
    rnd = int(os.path.basename(lines[-1])[1:3])
    if rnd >= 6:                                                     # round 6 on: the committed line IS the driver's compact line
        text = open(lines[-1]).read().strip()
        assert len(text) < 8000 and '\n' not in text
        d = _strict_loads(text)
        assert COMPACT_TOP <= set(d) and COMPACT_ROOFLINE <= set(d['roofline']) and COMPACT_CPU <= set(d['cpu_baseline']) and d['final'] is True
        assert d['metric'] == 'frames/sec GMM-score+fwd-bwd, 39-d MFCC, 2048-mix' and d['unit'] == 'frames/s' and d['vs_baseline'] is None
        assert d['value'] == pytest.approx(d['config']['frames_per_step_total'] / (d['ms_per_step'] * 1e-3), rel=1e-5)
        r = d['roofline']
        assert r['frac'] == pytest.approx(r['achieved'] / r['peak'], rel=1e-5) and r['frac_of_f16_dense_peak'] == pytest.approx(r['achieved'] / 2516.6, rel=1e-5)
        assert r['kernel_avg_ms'] <= d['ms_per_step'] * 1.02
        assert r['traffic'] is not None and r['kernel_code_sha16'], 'the line was printed without a PMC traffic figure'
        summ = open(os.path.join(ROOT, 'profiles', 'r%02d_bench_summary.txt' % rnd)).read()
        assert r['kernel_code_sha16'] in summ and r['traffic_file'] == 'profiles/r%02d_bench_summary.txt' % rnd
        for k in ('value_em2_model', 'value_em3_model', 'estep_ms', 'c4_ms_per_iteration', 'c5_frames_per_s', 'sustained_value', 'strict_f32_value'):
            assert r[k] is not None and r[k] > 0, k
        assert len(r['c4_em_iteration_ms']) == 3 and max(r['c4_em_iteration_ms']) <= 1.3 * r['c4_em_iteration_ms'][0]      # VERDICT r5 next #3, in the record
        c = d['cpu_baseline']
        assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] == pytest.approx(max(c['gemm_value'], c['vectorised_value']))
        full = json.load(open(os.path.join(ROOT, 'profiles', 'r%02d_bench_full.json' % rnd)))      # ... and the full record beside it is the same run
        assert full['value'] == pytest.approx(d['value'], rel=1e-6) and full['roofline']['kernel_avg_ms'] == pytest.approx(r['kernel_avg_ms'], rel=1e-6)
        return
    _validate_line(d)
    if rnd >= 5:                                                     # round 5: the whole record is inside `roofline`
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


# ------------------------------------------------------------------------------------------------
# round 6: the line the driver reads is a COMPACT record (BENCH_r05.parsed was null: the 22.8-KB line outgrew the reader)
# ------------------------------------------------------------------------------------------------
COMPACT_TOP = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
               'config', 'roofline', 'cpu_baseline', 'gpu_over_cpu', 'full_record', 'final'}
COMPACT_ROOFLINE = {'bound', 'achieved', 'peak', 'unit', 'frac', 'frac_denominator', 'frac_of_f16_dense_peak', 'frac_executed', 'traffic',
                    'traffic_over_algorithmic', 'kernel', 'kernel_avg_ms', 'kernel_code_sha16', 'sustained_value', 'fresh_batches_value',
                    'pcie_inclusive_value', 'strict_f32_value', 'strict_f32_frac', 'estep_ms', 'c4_ms_per_iteration', 'c5_frames_per_s',
                    'value_em2_model', 'value_em3_model', 'off_pipe_mixture_share_em2', 'off_pipe_mixture_share_em3'}
COMPACT_CPU = {'value', 'unit', 'cores', 'kind', 'sample', 'value_leg', 'gemm_value', 'vectorised_value', 'faithful_value'}


def _strict_loads(text):
    def no_constants(name):
        raise AssertionError('non-strict JSON constant %s in the line' % name)
    return json.loads(text, parse_constant=no_constants)


def _no_prose(o, limit, path=''):
    """every string of the compact record is a name, not a paragraph"""
    if isinstance(o, dict):
        for k, v in o.items():
            _no_prose(v, limit, path + '/' + k)
    elif isinstance(o, list):
        for v in o:
            _no_prose(v, limit, path)
    elif isinstance(o, str):
        assert len(o) <= limit, (path, len(o))


def test_compact_line_from_the_round5_record_fits_the_driver_and_keeps_the_contract():
    full = json.load(open(os.path.join(ROOT, 'profiles', 'r05_bench_line.json')))        # the record the driver could NOT parse
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full)
    assert len(line) <= bench.COMPACT_TARGET_CHARS < bench.COMPACT_MAX_CHARS < 8000 and '\n' not in line
    d = _strict_loads(line)
    assert COMPACT_TOP <= set(d) and COMPACT_ROOFLINE <= set(d['roofline']) and COMPACT_CPU <= set(d['cpu_baseline'])
    assert 'workload' in d['config'] and 'model' not in d['config']
    _no_prose(d, 340)
    # the numbers are the full record's (7 significant digits)
    assert d['value'] == pytest.approx(full['value'], rel=1e-6) and d['ms_per_step'] == pytest.approx(full['ms_per_step'], rel=1e-6)
    r, rf = d['roofline'], full['roofline']
    for k in ('achieved', 'peak', 'frac', 'traffic', 'kernel_avg_ms', 'sustained_value', 'strict_f32_value'):
        assert r[k] == pytest.approx(rf[k], rel=1e-6), k
    assert r['frac'] == pytest.approx(r['achieved'] / r['peak'], rel=1e-5)
    assert r['kernel_code_sha16'] == rf['traffic_raw']['kernel_code_sha16'] and r['kernel'] == rf['kernel']
    assert r['estep_ms'] == pytest.approx(full['extra']['estep_ms'], rel=1e-6)
    assert r['c4_ms_per_iteration'] == pytest.approx(full['extra']['configs']['C4']['ms_per_iteration'], rel=1e-6)
    assert r['c5_frames_per_s'] == pytest.approx(full['extra']['configs']['C5']['value'], rel=1e-6)
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] == full['cpu_baseline']['cores'] and c['value'] == pytest.approx(max(c['gemm_value'], c['vectorised_value']))
    assert d['gpu_over_cpu'] == pytest.approx(full['value'] / full['cpu_baseline']['value'], rel=1e-5)
    assert d['value'] == pytest.approx(d['config']['frames_per_step_total'] / (d['ms_per_step'] * 1e-3), rel=1e-5)


def test_compact_line_is_strict_json_whatever_the_extras_left_behind():
    full = json.load(open(os.path.join(ROOT, 'profiles', 'r05_bench_line.json')))
    full['roofline']['sustained_value'] = float('nan')             # a side measurement that divided by zero
    full['roofline']['traffic'] = float('inf')
    full['extra']['estep_ms'] = np.float32(48.5)                   # NumPy scalars straight from a kernel timer
    full['extra']['configs']['C4']['em_iterations'] = [dict(ms=np.float64(360.0), mixtures_off_the_matrix_pipe=0.0),
                                                        dict(ms=float('nan'), mixtures_off_the_matrix_pipe=np.float64(0.028))]
    full['extra']['error'] = 'RuntimeError: ' + 'x' * 5000          # an exception message of any length
    full['config']['workload'] = 'w' * 4000
    d = _strict_loads(bench.compact_line(full))
    assert d['roofline']['sustained_value'] is None and d['roofline']['traffic'] is None and d['roofline']['estep_ms'] == 48.5
    assert d['roofline']['c4_em_iteration_ms'] == [360.0, None] and len(d['extra_error']) <= 200 and len(d['config']['workload']) <= 330
    # an early line (no extras yet, no CPU leg on ranks > 0) is just as valid
    early = {k: v for k, v in full.items() if k not in ('extra', 'roofline_estep', 'cpu_baseline')}
    e = _strict_loads(bench.compact_line(dict(early, final=False)))
    assert e['final'] is False and e['cpu_baseline'] is None and e['roofline']['estep_ms'] is None and e['value'] == pytest.approx(full['value'], rel=1e-6)


def test_full_record_goes_to_a_file_not_to_stdout(tmp_path, monkeypatch, capsys):
    full = json.load(open(os.path.join(ROOT, 'profiles', 'r05_bench_line.json')))
    full['roofline']['sustained_value'] = float('nan')
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    bench.emit(full, final=True)
    out = capsys.readouterr().out
    assert out.count('\n') == 1 and len(out) < 8000                    # ONE line, and it is the compact one
    d = _strict_loads(out)
    assert d['final'] is True and d['full_record'] == bench.FULL_RECORD_NAME
    for where in (tmp_path / bench.FULL_RECORD_NAME, tmp_path / 'gpurun_out' / bench.FULL_RECORD_NAME):
        whole = _strict_loads(where.read_text())
        assert whole['extra']['configs']['C4']['ms_per_iteration'] == full['extra']['configs']['C4']['ms_per_iteration']
        assert whole['roofline']['sustained_value'] is None and whole['roofline']['note'] == full['roofline']['note']
