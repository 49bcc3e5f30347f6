"""The driver's multi-GPU command, rehearsed on the one-GPU box.

`python bench.py --gpus N` is what the driver launches at N = 2, 4, 8 (as `python -m torch.distributed.run ... bench.py --gpus N`
or bare, when bench.py spawns the ranks itself).  No 8-GPU node is available to the builder, so these tests run that exact
command with POCCALA_SHARE_DEVICE=1 -- N rank processes on device 0, the host transport in place of RCCL (which refuses two
ranks on one device) -- on a REDUCED problem (--utts / --mix: the control flow is what is rehearsed, not the timing) and hold:
one JSON line, n_gpus, the transport, the per-rank exchange block, value = frames of all ranks / the slowest rank's time, a
non-zero exit when a rank dies (the parent terminates the others), the watchdog when a communicator never comes up, and the
strong-scaling line of config 4 (`--workload C4 --gpus N`).  NO N > 1 TIMING EXISTS: nothing here is a scaling measurement.

The module name sorts before the other GPU modules on purpose: its subprocesses start before this process has made a context.
Every test is a subprocess; this process never touches the GPU here.
"""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ['--utts', '32', '--mix', '256', '--steps', '2', '--warmup', '1', '--sustain', '0', '--cpu-baseline', '0']


def _env(**extra):
    env = dict(os.environ, POCCALA_SHARE_DEVICE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'POCCALA_CTRL_TOKEN'):
        env.pop(k, None)
    env.update(extra)
    return env


def _run(cmd, env, timeout):
    t0 = time.time()
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    return p.returncode, p.stdout, p.stderr, time.time() - t0


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith('{')]


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _full_record():
    """the whole record rank 0 wrote beside bench.py (the stdout line is the compact one since round 6)"""
    return json.load(open(os.path.join(ROOT, 'bench_full.json')))


def _final_line(out):
    """bench.py prints the compact line as soon as the headline is final (final: false) and again at the end (final: true): the
    driver reads the LAST one.  Every line must fit the driver's stdout tail."""
    lines = _json_lines(out)
    assert 1 <= len(lines) <= 2 and lines[-1]['final'] is True and all(len(json.dumps(l)) < 8000 for l in lines), out[-3000:]
    if len(lines) == 2:
        assert lines[0]['final'] is False and lines[0]['value'] == lines[1]['value'] and lines[0]['ms_per_step'] == lines[1]['ms_per_step']
    return lines[-1]


def _check_line(d, n):
    assert d['metric'].startswith('frames/sec GMM-score+fwd-bwd')
    assert d['n_gpus'] == n and d['scaling'] == 'weak' and d['unit'] == 'frames/s'
    assert d['config']['transport'] == 'host-rehearsal'          # (RCCL refuses two ranks on one device; on a node: 'rccl', rccl_nranks == n)
    assert d['config']['workload'].startswith('REDUCED')
    # value = the frames ALL ranks processed / the slowest rank's time for exactly `steps` steps
    total = d['config']['frames_per_step_total']
    assert total == n * 32 * 300
    assert d['value'] == pytest.approx(total * d['steps'] / (d['ms_per_step'] * 1e-3 * d['steps']), rel=1e-5)
    assert 'extra_error' not in d, d.get('extra_error')
    r = d['roofline']
    assert r['kernel_avg_ms'] > 0 and d['cpu_baseline'] is None      # (the CPU leg runs at N = 1 only)
    assert r['estep_ms'] > 0 and r['exchange_ms'] >= 0 and r['reduce_scatter_ms'] >= 0 and r['all_gather_ms'] >= 0 and r['exchange_payload'] == 'f32'
    full = _full_record()
    assert full['value'] == pytest.approx(d['value'], rel=1e-6) and full['final'] is True
    ex = full['extra']
    assert 'error' not in ex, ex.get('error')
    per = ex['exchange']['per_rank']
    assert sorted(r_['rank'] for r_ in per) == list(range(n))
    for r_ in per:
        assert r_['estep_ms'] > 0 and r_['exchange_ms'] >= 0 and r_['reduce_scatter_ms'] >= 0 and r_['all_gather_ms'] >= 0
    assert r['exchange_ms'] == pytest.approx(max(r_['exchange_ms'] for r_ in per), rel=1e-6)
    assert ex['exchange']['payload'] == 'f32' and ex['exchange']['wire']['world'] == n
    assert ex['estep_pipelined'].get('error') is None and len(ex['estep_pipelined']['per_rank']) == n


def test_bench_self_spawned_two_ranks():
    """`python bench.py --gpus 2` with no launcher variables: bench.py spawns its rank processes (before anything touches HIP)."""
    rc, out, err, dt = _run([sys.executable, 'bench.py', '--gpus', '2'] + SMALL, _env(), 600)
    assert rc == 0, err[-2000:]
    _check_line(_final_line(out), 2)


def test_bench_under_the_drivers_launcher_four_ranks():
    """The driver's form, word for word: python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 4 --steps K --warmup W (the launcher imports torch; the ranks never do)."""
    import importlib.util
    if importlib.util.find_spec('torch') is None:      # (found, NOT imported: torch's wheel bundles a second HIP runtime, and two of them in
        pytest.skip('torch is not installed')          #  this process end in a double free at interpreter exit -- the launcher runs in a child)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '4', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), 'bench.py', '--gpus', '4'] + SMALL
    rc, out, err, dt = _run(cmd, _env(), 900)
    assert rc == 0, err[-2000:]
    _check_line(_final_line(out), 4)


def test_a_dying_rank_fails_the_job():
    """Rank 1 dies right behind the timed loop (POCCALA_TEST_DIE_RANK): the parent must terminate the rank left waiting at the next
    barrier and exit non-zero -- in bounded time, not at the watchdog's."""
    rc, out, err, dt = _run([sys.executable, 'bench.py', '--gpus', '2', '--extra-timeout', '300'] + SMALL, _env(POCCALA_TEST_DIE_RANK='1'), 600)
    assert rc != 0
    assert dt < 240, dt
    # (rank 0 may still print its line -- the timed number was final before the death -- but then only with the failure inside `extra`)
    lines = _json_lines(out)
    assert len(lines) <= 2 and all('extra_error' in l for l in lines if l['final'])


def test_the_watchdog_fires_when_the_communicator_never_comes_up():
    """POCCALA_TEST_HANG_COMM: every rank sleeps where pcl_comm_init would be.  The timed number is final by then: rank 0 prints the
    line WITHOUT the extras after --extra-timeout seconds and the job exits non-zero (3)."""
    rc, out, err, dt = _run([sys.executable, 'bench.py', '--gpus', '2', '--extra-timeout', '4'] + SMALL, _env(POCCALA_TEST_HANG_COMM='1'), 600)
    assert rc == 3, (rc, err[-2000:])
    lines = _json_lines(out)
    assert len(lines) == 2 and not lines[0]['final'] and not lines[1]['final']        # the early line, and the watchdog's
    d = lines[-1]
    assert d['n_gpus'] == 2 and d['value'] > 0 and d['value'] == lines[0]['value']
    assert 'did not finish' in d['extra_error']


def test_bench_config4_strong_scaling_line_two_ranks():
    """`--workload C4 --gpus 2`: the corpus' batches dealt round-robin to the ranks, E-step + exchange + M-step timed per EM
    iteration (the configuration north_star names for the collective), batches created inside the timed region."""
    cmd = [sys.executable, 'bench.py', '--gpus', '2', '--workload', 'C4', '--c4-batches', '4', '--utts', '48', '--mix', '256', '--units', '200',
           '--steps', '1', '--warmup', '1', '--iters', '2']
    rc, out, err, dt = _run(cmd, _env(), 900)
    assert rc == 0, err[-2000:]
    lines = _json_lines(out)
    assert len(lines) == 1, out
    d = lines[0]
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong'
    assert d['config']['transport'] == 'host-rehearsal' and d['config']['workload'].startswith('REDUCED')
    r = d['detail']
    assert r['batches_on_this_rank'] == 2 and r['n_gpus'] == 2
    nfr = 4 * 48 * 300
    assert d['value'] == pytest.approx(nfr / (d['ms_per_step'] * 1e-3), rel=1e-9)
    assert r['fresh_batches']['statistics_same_bits_as_resident'] is True
    assert r['fresh_batches']['statistics_same_bits_after_an_em_iteration'] is True
    assert len(r['em_iterations']) == 2 and all(e['ms'] > 0 for e in r['em_iterations'])
    assert r['kernel_ms_per_iteration_rank0']['reduce_scatter'] >= 0 and r['kernel_ms_per_iteration_rank0']['all_gather'] >= 0
