#!/usr/bin/env python3
"""bench.py -- frames/sec of GMM-score + forward-backward (BASELINE.json metric) on N MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic utterances: state-major GMM scoring
of every (frame, label state) pair of the batch followed by the Baum-Welch forward/backward pass loop
(LHMM.baulm_welch: alpha, beta, xi, gamma, pi, per-frame posteriors).  The default workload is the
per-GPU shard of BASELINE config 4 (the configuration the metric is quoted on: 39-dim MFCC, 2048-mix,
3000 tied states; 8 GPUs x 1024 utterances = the full 8192-utterance E-step), so scaling is weak:
every rank owns its own 1024 utterances and the path needs no collective (the statistics all-reduce
belongs to the accumulate stage, measured separately under "extra").

Inputs (frames, model, batch descriptors) are resident in HBM before the timed region.  Timing:
barrier + device sync on both sides of exactly K steps, MAX over ranks; rank 0 prints ONE JSON line.
Multi-GPU: one process per GPU.  Launched by torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the
env), or by itself: `python bench.py --gpus N` with no launcher env spawns N rank processes (subprocess, before
anything touches HIP; the parent never does), waits for them and exits non-zero if any of them failed.  The barrier /
max / RCCL-id broadcast go over a small authenticated TCP control plane (poccala_amd.distributed.Control) so that the
GPU processes never import torch, whose wheel bundles a second HIP runtime; the GPU work goes through
libpoccala_hip.so and RCCL.
"""
import argparse
import json
import os
import secrets
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

T_PROCESS_START = time.perf_counter()
CPU_SAMPLE_FRAMES = 60            # frames per utterance in the CPU baseline sample (a fifth of an utterance per core: ~20 s of the cache-blocked vectorised leg with every core busy)
CPU_GEMM_FRAMES = 300             # ... and in the GEMM leg's: the whole utterance (seconds per core)
CPU_FAITHFUL_ROWS = 12            # label states (of 60) the reference's per-mixture loop nest is timed on, scaled to all of them
FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 vector == FP32 matrix (v_mfma_f32_*_f32)
BF16_MFMA_PEAK_TFLOPS = 2516.6    # 256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz (MI355X_MICROARCH.md: ~2.5 PF dense)
HBM_PEAK_GBS = 8000.0
# what `dtype` says for the default f32-class path: the arithmetic, not a precision claim
C5_CORPUS_UTTS = 3336         # config 5: 1M frames = 3336 utterances x 300 frames
C5_CHUNK_UTTS = 417          # utterances per stream chunk (8 chunks; profiles/r04_c5_chunk_sweep.txt: 139 -> 0.76, 417 -> 0.80, 834 -> 0.82, 1668 -> 0.76 M frames/s)
DTYPE_NAME = 'f32-class (f16x2 split operands, f32 accumulate; f64 dynamic programming)'


def resolve_payload(args, world):
    """wire format of the E-step exchange: f32 by default as soon as there is a wire (1.94 GB per exchange
    instead of 3.9), f64 on one rank where nothing travels; --payload overrides."""
    from poccala_amd import PCL_F32, PCL_F64
    name = args.payload if args.payload != 'auto' else ('f32' if world > 1 else 'f64')
    return PCL_F32 if name == 'f32' else PCL_F64


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=5)
    p.add_argument('--warmup', type=int, default=1)
    p.add_argument('--workload', default='C4shard', help='poccala_amd.synth.CONFIGS key')
    p.add_argument('--utts', type=int, default=0, help='override utterances per GPU (--workload C4: per batch)')
    p.add_argument('--mix', type=int, default=0, help='override mixtures per state (rehearsals of the control flow on a small model; not the named configuration)')
    p.add_argument('--units', type=int, default=0, help='override the number of units (rehearsals)')
    p.add_argument('--c4-batches', type=int, default=0, help='--workload C4: batches of the corpus (default 8 = config 4; rehearsals use fewer)')
    p.add_argument('--precision', default='f32', choices=['f32', 'f64'])
    p.add_argument('--batches', type=int, default=2,
                   help='resident utterance batches (each of the full per-GPU size) that successive steps alternate between, as a '
                        'data loader would: the forward-backward of step k runs on the library\'s second stream beside the '
                        'scoring of step k+1; 1 = every step re-scores the same batch and the two phases serialise')
    p.add_argument('--cpu-baseline', type=int, default=1, help='0 = skip the CPU baseline leg')
    p.add_argument('--words', type=int, default=20000, help='--workload C5shard: random words added to the synthetic lexicon')
    p.add_argument('--max-tokens', type=int, default=8192, help='--workload C5shard: live tokens per utterance')
    p.add_argument('--c5-chunk', type=int, default=C5_CHUNK_UTTS, help='--workload C5: utterances per stream chunk (the 3336-utterance corpus is cut into 3336 / this many chunks)')
    p.add_argument('--extra', type=int, default=1, help='0 = skip the untimed extra measurements')
    p.add_argument('--sustain', type=float, default=10.0, help='seconds the headline loop is held for value_sustained (0 = skip it and the PCIe-inclusive loop)')
    p.add_argument('--extra-timeout', type=int, default=600, help='seconds the untimed extras (and the shutdown) may take before rank 0 prints the line without them')
    p.add_argument('--traffic-bytes', type=float, default=None,
                   help='HBM bytes per scoring launch from a separate rocprofv3 --pmc pass, corrected as MI355X_MICROARCH.md '
                        'prescribes (2 x FETCH_SIZE for the wide streaming reads + WRITE_SIZE); default: the figure committed in '
                        'profiles/ (tools/gpu_profile.sh)')
    p.add_argument('--iters', type=int, default=0, help='--workload C4: EM iterations run IN SEQUENCE (each on the model the previous M-step left), reported per iteration')
    p.add_argument('--c-covariance', type=float, default=1e-3, help='--workload C4: the variance floor of GMM.update_param (the reference driver passes 1e-6: init.py:30 -> Controller.py:151)')
    p.add_argument('--payload', default='auto', choices=['auto', 'f64', 'f32'],
                   help='wire format of the E-step exchange: auto = f32 when there is more than one rank (half the bytes on xGMI), f64 on one')
    return p.parse_args()


# ------------------------------------------------------------------------------------------------
# `python bench.py --gpus N` without a launcher: spawn the N ranks ourselves.
# ------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    """The parent never imports poccala_amd (no HIP call here): it starts one child per rank with the launcher env
    torch.distributed.run would give it, passes the children's stdout through (rank 0 prints the JSON line), and
    returns non-zero if any child failed (the others are then terminated by PID)."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    token = secrets.token_hex(16)
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), POCCALA_CTRL_TOKEN=token)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    pending = list(procs)
    while pending:
        for p_ in list(pending):
            code = p_.poll()
            if code is None:
                continue
            pending.remove(p_)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in pending:          # a failed rank would leave the others waiting at a barrier
                    q.terminate()
        time.sleep(0.05)
    return rc


# ------------------------------------------------------------------------------------------------
# CPU baseline leg: the oracle (a port of the reference's arithmetic) timed on the host cores.
# ------------------------------------------------------------------------------------------------
_CPU_JOBS = None      # set by cpu_baseline() BEFORE the pool forks: the workers read their job by index, nothing is pickled


def _cpu_job(u, nfr=None):
    """(frames (T,D) f64, [(mean, var, w)] of the label's states, A, pi) of sample utterance u, from the forked globals."""
    from poccala_amd.engine import embedded_structure
    cfg, mean, var, w, trans, frames, lens, begin, labels = _CPU_JOBS
    e = 3
    lab = labels[u]
    x = frames[begin[u]:begin[u] + min(int(lens[u]), nfr or CPU_SAMPLE_FRAMES)].astype(np.float64)   # bounded sample
    gm = [(mean[i * e + k], var[i * e + k], w[i * e + k]) for i in lab for k in range(e)]
    a, pi = embedded_structure(len(lab), [trans[i] for i in lab])
    return x, gm, a, pi


def _cpu_vectorised_utt(u):
    """One utterance of score + forward-backward with the vectorised float64 oracle; returns its own elapsed time."""
    from threadpoolctl import threadpool_limits
    from oracle import poccala_oracle as po
    x, gmms_per_row, a, pi = _cpu_job(u)
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        rows = [np.zeros(x.shape[0])]
        for (mean, var, w) in gmms_per_row:
            rows.append(po.gmm_point_blocked(x, mean, var, w))      # the reference's arithmetic, evaluated in cache-sized blocks
        rows.append(np.full(x.shape[0], -np.inf))
        b = np.array(rows)
        po.baum_welch(a, pi, [b])
        return time.perf_counter() - t0, x.shape[0]


def _cpu_gemm_utt(u):
    """Informational third figure: the same log-likelihoods through the expanded quadratic form as ONE float64
    GEMM per state on the host BLAS (single thread per worker) + the vectorised forward-backward.  This is not
    the reference's arithmetic (it is the formulation the GPU kernel uses); it shows what an optimised CPU
    implementation of the same mathematics would do on these cores."""
    from threadpoolctl import threadpool_limits
    from oracle import poccala_oracle as po
    x, gmms_per_row, a, pi = _cpu_job(u, CPU_GEMM_FRAMES)      # the whole utterance: the per-state parameter preparation is paid once per utterance, as in a real job
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        rows = [np.zeros(x.shape[0])]
        for (mean, var, w) in gmms_per_row:
            rows.append(po.gmm_point_gemm(x, mean, var, w))
        rows.append(np.full(x.shape[0], -np.inf))
        po.baum_welch(a, pi, [np.array(rows)])
        return time.perf_counter() - t0, x.shape[0]


def _cpu_faithful_sample(u):
    """The reference's own loop nest (per frame x per mixture NumPy calls; per-(t,j) LSE) on a tiny
    sample: 1 frame of scoring for every row + one full forward/backward lattice.  Returns seconds per frame."""
    from threadpoolctl import threadpool_limits
    from oracle import poccala_oracle as po
    x, gmms_per_row, a, pi = _cpu_job(u)
    nf = 1
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        sub = gmms_per_row[::max(1, len(gmms_per_row) // CPU_FAITHFUL_ROWS)]
        for (mean, var, w) in sub:
            for t in range(nf):
                po.faithful_gmm_point(x[t], mean, var, w)
        t_score = (time.perf_counter() - t0) / nf * len(gmms_per_row) / len(sub)     # seconds per frame, scaled to all rows
        n, T = a.shape[0], x.shape[0]
        b = np.random.default_rng(0).standard_normal((n, T)) - 60.0
        b[0] = 0.0
        b[-1] = -np.inf
        t0 = time.perf_counter()
        po.faithful_forward_backward(a, pi, b)
        t_fb = 3 * (time.perf_counter() - t0) / T                    # 3 passes (quirk Q6), seconds per frame
    return t_score + t_fb


def cpu_baseline(cfg, mean, var, w, trans, frames, lens, begin, labels):
    """The oracle timed on the host cores: one sample utterance per core.  The job data sits in a module global BEFORE the
    pool forks, the workers receive an index and time themselves; a leg's rate = frames of the sample / the slowest worker's
    time (what the job would take with every core busy), so neither pickling nor the pool's dispatch is in the figure."""
    import multiprocessing as mp
    global _CPU_JOBS
    cores = os.cpu_count() or 1
    n_utt = min(cores, len(labels))
    _CPU_JOBS = (cfg, mean, var, w, trans, frames, lens, begin, labels)
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    t_leg = {}
    with mp.get_context('fork').Pool(n_utt) as pool:
        idx = list(range(n_utt))
        t0 = time.perf_counter()
        vec = pool.map(_cpu_vectorised_utt, idx, chunksize=1)
        t_leg['vectorised'] = time.perf_counter() - t0
        t0 = time.perf_counter()
        per_frame = pool.map(_cpu_faithful_sample, idx, chunksize=1)
        t_leg['faithful'] = time.perf_counter() - t0
        t0 = time.perf_counter()
        gemm = pool.map(_cpu_gemm_utt, idx, chunksize=1)
        t_leg['gemm'] = time.perf_counter() - t0
    _CPU_JOBS = None
    frames_done = int(sum(n for _, n in vec))
    nfr, nrows = vec[0][1], 3 * len(labels[0])
    faithful = n_utt / float(max(per_frame))                       # every core one frame at a time, the slowest core sets the rate
    vec_value, gemm_value = frames_done / max(t for t, _ in vec), int(sum(n for _, n in gemm)) / max(t for t, _ in gemm)
    lead = 'gemm' if gemm_value >= vec_value else 'vectorised'
    common = ('%d utterances, one per core (forked workers reading the job from inherited memory, timed inside the workers: frames / slowest worker): '
              'the first %d frames of each in the vectorised leg, the first %d (the whole utterance) in the GEMM leg; score %d label states x %d '
              'mixtures + 3-pass forward-backward' % (n_utt, nfr, gemm[0][1], nrows, cfg['M']))
    legs = dict(gemm='oracle.gmm_point_gemm: the expanded quadratic form as one float64 BLAS GEMM per state, 1 thread per worker -- the STRONGEST CPU formulation '
                     'of this path (what an optimised CPU implementation would do; not the reference\'s order of operations)',
                vectorised='oracle.gmm_point_blocked: the reference\'s arithmetic (subtract, scale, square, sum; util.py:22-31) vectorised in float64 NumPy and '
                           'evaluated in cache-sized blocks of 16 frames x 128 mixtures',
                faithful='the reference as written: its loop nest (per frame x per mixture NumPy calls, per-(t,j) LSE): 1 frame x %d of the %d label states '
                         'of scoring (scaled to all of them) + one faithful forward/backward lattice of %d frames per core, frames/s = cores / slowest '
                         'core\'s seconds per frame' % (len(labels[0]) * 3 // max(1, len(labels[0]) * 3 // CPU_FAITHFUL_ROWS), nrows, nfr))
    return dict(value=max(vec_value, gemm_value), unit='frames/s', cores=n_utt, kind='port', ipc_excluded=True, value_leg=lead,
                sample_short='%d utterances, one per core, timed inside forked workers (frames / slowest worker): %d frames each (%s leg = value; vectorised leg %d frames); %d label states x %d mixtures + 3-pass forward-backward' % (n_utt, gemm[0][1] if lead == 'gemm' else nfr, lead, nfr, nrows, cfg['M']),
                sample=common + '; value = the strongest of the CPU legs: ' + legs[lead],
                vectorised_value=vec_value, vectorised_sample=legs['vectorised'],
                gemm_value=gemm_value, gemm_sample=legs['gemm'],
                faithful_value=faithful, faithful_sample=legs['faithful'],
                worker_s=dict(vectorised_max=max(t for t, _ in vec), vectorised_mean=float(np.mean([t for t, _ in vec])),
                              gemm_max=max(t for t, _ in gemm), gemm_mean=float(np.mean([t for t, _ in gemm])),
                              faithful_s_per_frame_max=float(max(per_frame)), faithful_s_per_frame_mean=float(np.mean(per_frame))),
                leg_wall_s=t_leg)


# ------------------------------------------------------------------------------------------------
# The line the driver reads.  Round 5's record had grown to 22.8 KB and the driver could not parse it (BENCH_r05.parsed == null):
# the LAST stdout line is now a compact record of numbers (target <= 4 KB, never >= 8000 characters -- the driver's stdout tail),
# strict JSON (no NaN / Infinity); the full record with its prose goes to bench_full.json next to this file and under gpurun_out/.
# ------------------------------------------------------------------------------------------------
COMPACT_TARGET_CHARS = 4096
COMPACT_MAX_CHARS = 7900
FULL_RECORD_NAME = 'bench_full.json'


def _finite(o, sig=7):
    """the same object with floats rounded to `sig` significant digits, non-finite floats and NumPy scalars mapped to JSON types"""
    import math
    if isinstance(o, dict):
        return {str(k): _finite(v, sig) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v, sig) for v in o]
    if isinstance(o, (bool, np.bool_)):
        return bool(o)
    if isinstance(o, (int, np.integer)):
        return int(o)
    if isinstance(o, (float, np.floating)):
        f = float(o)
        if not math.isfinite(f):
            return None
        return float('%.*g' % (sig, f)) if sig else f
    return o


def _pick(d, *path):
    for k in path:
        if not isinstance(d, dict):
            return None
        d = d.get(k)
    return d


def compact_record(full):
    """The driver's line from the full record: the contract's keys, `config`, `roofline` and `cpu_baseline` as numbers and short
    names -- no prose.  Pure (tests/test_bench_schema.py builds it from committed full records)."""
    cfg, rf, cpu, ex = full.get('config') or {}, full.get('roofline') or {}, full.get('cpu_baseline'), full.get('extra') or {}
    workload = str(cfg.get('workload', ''))
    out = {k: full.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                     'vs_baseline', 'dtype', 'data')}
    out['config'] = {'workload': workload if len(workload) <= 330 else workload[:327] + '...'}
    for k in ('utterances_total', 'frames_per_step_total', 'resident_batches', 'transport', 'rccl_nranks', 'device', 'cus'):
        out['config'][k] = cfg.get(k)
    r = {k: rf.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac')}
    den = str(rf.get('frac_denominator') or '')
    r['frac_denominator'] = den.split(':')[0][:96] if den else None
    for k in ('frac_of_f16_dense_peak', 'frac_executed', 'traffic', 'traffic_over_algorithmic', 'kernel', 'kernel_avg_ms', 'launches',
              'flop_per_launch', 'hbm_algorithmic_bytes_per_launch', 'fb_kernel_avg_ms', 'fb_kernel_alone_ms'):
        r[k] = rf.get(k)
    r['kernel_code_sha16'] = _pick(rf, 'traffic_raw', 'kernel_code_sha16')
    r['traffic_file'] = _pick(rf, 'traffic_raw', 'file')
    for k in ('sustained_value', 'fresh_batches_value', 'pcie_inclusive_value', 'strict_f32_value', 'strict_f32_frac',
              'value_em2_model', 'value_em3_model', 'off_pipe_mixture_share_em2', 'off_pipe_mixture_share_em3'):
        r[k] = rf.get(k)
    for k in ('score_kernel_ms_em3', 'coarse_kernel_ms_em3'):
        r[k] = _pick(ex, 'em_shaped_models', k)
    r['estep_ms'] = ex.get('estep_ms')
    per = _pick(ex, 'exchange', 'per_rank') or []              # N > 1: what the statistics exchange cost (max over ranks)
    for k in ('exchange_ms', 'reduce_scatter_ms', 'mstep_owned_ms', 'all_gather_ms'):
        r[k] = max([p_.get(k) or 0.0 for p_ in per]) if per else None
    r['exchange_payload'] = _pick(ex, 'exchange', 'payload')
    r['estep_pipelined_ms'] = _pick(ex, 'estep_pipelined', 'estep_ms')
    r['accumulate_ms'] = ex.get('accumulate_ms')
    r['estep_frac_executed'] = _pick(full, 'roofline_estep', 'frac_executed')
    r['c4_ms_per_iteration'] = _pick(ex, 'configs', 'C4', 'ms_per_iteration')
    r['c4_fresh_ms_per_iteration'] = _pick(ex, 'configs', 'C4', 'fresh_batches', 'ms_per_iteration')
    em = _pick(ex, 'configs', 'C4', 'em_iterations') or []
    r['c4_em_iteration_ms'] = [e.get('ms') for e in em] or None
    r['c4_em_off_pipe_share'] = [e.get('mixtures_off_the_matrix_pipe') for e in em] or None
    r['c2_frames_per_s'] = _pick(ex, 'configs', 'C2', 'value')
    r['c3_frames_per_s'] = _pick(ex, 'configs', 'C3', 'value')
    r['c5_frames_per_s'] = _pick(ex, 'configs', 'C5', 'value')
    r['c5_shard_frames_per_s'] = _pick(ex, 'configs', 'C5shard', 'value')
    r['c5_decode_kernel_ms'] = _pick(ex, 'configs', 'C5shard', 'decode_kernel_ms')
    out['roofline'] = r
    if cpu:
        c = {k: cpu.get(k) for k in ('value', 'unit', 'cores', 'kind', 'value_leg')}
        c['sample'] = cpu.get('sample_short') or str(cpu.get('sample', ''))[:200]
        for k in ('gemm_value', 'vectorised_value', 'faithful_value'):
            c[k] = cpu.get(k)
        c['wall_s'] = sum((cpu.get('leg_wall_s') or {}).values()) or None
        out['cpu_baseline'] = c
        out['gpu_over_cpu'] = (full['value'] / cpu['value']) if cpu.get('value') and full.get('value') else None
    else:
        out['cpu_baseline'] = None
    err = ex.get('error') if isinstance(ex, dict) else None
    if err:
        out['extra_error'] = str(err)[:200]
    out['full_record'] = full.get('full_record')
    out['final'] = bool(full.get('final', True))
    return _finite(out)


def compact_line(full):
    """compact_record as ONE strict-JSON line that fits the driver's stdout tail: optional keys are dropped (last first) rather than
    the line lost, should it ever outgrow the limit."""
    rec = compact_record(full)
    line = json.dumps(rec, allow_nan=False, separators=(', ', ': '))
    optional = ['estep_pipelined_ms', 'mstep_owned_ms', 'coarse_kernel_ms_em3', 'score_kernel_ms_em3', 'c5_decode_kernel_ms', 'c5_shard_frames_per_s', 'c3_frames_per_s', 'c2_frames_per_s', 'c4_em_off_pipe_share', 'c4_em_iteration_ms',
                'c4_fresh_ms_per_iteration', 'estep_frac_executed', 'accumulate_ms', 'fb_kernel_alone_ms', 'fb_kernel_avg_ms',
                'hbm_algorithmic_bytes_per_launch', 'flop_per_launch', 'launches', 'traffic_file', 'frac_denominator']
    while len(line) > COMPACT_TARGET_CHARS and optional:
        rec['roofline'].pop(optional.pop(0), None)
        line = json.dumps(rec, allow_nan=False, separators=(', ', ': '))
    if len(line) >= COMPACT_MAX_CHARS:                     # cannot happen with the key set above; never lose the contract's keys to it
        rec['config']['workload'] = rec['config']['workload'][:120]
        rec.pop('extra_error', None)
        line = json.dumps(rec, allow_nan=False, separators=(', ', ': '))
    assert len(line) < COMPACT_MAX_CHARS and '\n' not in line, len(line)
    return line


def write_full_record(full):
    """the whole record (prose, side measurements) as strict JSON: bench_full.json beside bench.py and under gpurun_out/ (merged back
    from a GPU box).  Returns the path written first, or None -- a read-only tree must not cost the line."""
    text = json.dumps(_finite(full, sig=0), allow_nan=False, indent=1)
    first = None
    for d in (ROOT, os.path.join(ROOT, 'gpurun_out')):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, FULL_RECORD_NAME), 'w') as f:
                f.write(text)
            first = first or os.path.join(os.path.basename(d) if d != ROOT else '', FULL_RECORD_NAME).lstrip('/')
        except OSError:
            pass
    return first


def emit(full, final):
    """rank 0: write the full record, print the compact line (and flush): called once as soon as the headline, the roofline and the CPU
    leg are final, and again at the end with what the untimed extras added -- a late failure leaves the early line as the last one."""
    full = dict(full, final=final)
    full['full_record'] = write_full_record(full)
    print(compact_line(full))
    sys.stdout.flush()


# per scoring kernel: name, the peak its arithmetic is priced against, and what that peak means
KERNELS = {
    0: ('gmm_score_kernel<39,2,32,double>', FP32_VECTOR_PEAK_TFLOPS / 2, 'f64 parity mode on the VALU (78.6 TFLOP/s f64 vector peak)'),
    1: ('gmm_score_kernel<39,3,64,float>', FP32_VECTOR_PEAK_TFLOPS, 'f32 VALU kernel (PCL_SCORE_VARIANT=1); peak = 157.3 TFLOP/s f32 vector'),
    3: ('gmm_score_mfma_kernel<39,2>', FP32_VECTOR_PEAK_TFLOPS,
        'quadratic form on the f32-input matrix pipe (v_mfma_f32_32x32x2_f32, PCL_SCORE_VARIANT=3); peak = 157.3 TFLOP/s '
        'dense f32 MFMA (= f32 vector peak)'),
    7: ('gmm_score_split16_kernel<39,2>', BF16_MFMA_PEAK_TFLOPS / 3,
        'quadratic form of the diagonal Gaussians as an f32-class contraction on the f16 matrix pipe: every f32 operand is '
        'scaled by an exact power of two per (state, feature) and written as the sum of two f16 pieces (22 significand '
        'bits), three of the four cross products are kept, f32 accumulate; the constant (relative to a per-state K0 added '
        'back in f64), the log-sum-exp reference and log-zero ride in the spare K slot of the f16 passes; tiles whose scaled '
        'features leave the f16 range are rescored by the direct-form kernel in the same call.  Measured |d ln b| vs float64 '
        '1.5e-5 at |ln b| ~ 85 (f32 FMA chain 1.0e-5), same parity tolerances.  achieved = ALGORITHMIC flops M(3D+4) per '
        '(frame,state) pair / kernel time; peak = 2516.6 TFLOP/s dense f16 MFMA / 3 products per f32-class product = 838.9 '
        '(f32-input MFMA peak: 157.3); the kernel executes 15 MFMAs of 32x32x16 per 1024 Gaussians = 480 MFMA flops per '
        'Gaussian: see executed_mfma_tflops.  On random operands the chip holds ~1.7-1.8 GHz under this kernel (2.4 GHz spec), '
        'matrix pipe 60-67 % busy (profiles/)'),
}


def make_roofline(score_variant, score_avg_ms, launches, scored_pairs, label_pairs, M, D, alg_bytes, traffic, traffic_raw, fb_avg_ms, dp_alone_ms):
    """The `roofline` object of the line, from the measured average launch time of the dominant kernel (HIP events on the library's
    stream inside the timed region) and the ALGORITHMIC work of a launch: scored (frame, state) pairs x M x (3D + 4) flop.
    Every fraction a reader may want is here, each against a named denominator: `frac` = achieved / peak where `peak` is what
    `frac_denominator` says; `frac_of_f16_dense_peak` = the same algorithmic rate against the guide's raw dense f16 MFMA peak;
    `frac_executed` = the MFMA flops the kernel actually issues (480 per Gaussian for 121 algorithmic) against that raw peak.
    The caller adds the loop-level figures (sustained, strict f32, PCIe inclusive, fresh batches) before the line is printed."""
    name, peak, note = KERNELS.get(score_variant, KERNELS[1])
    flop = scored_pairs * M * (3 * D + 4)
    achieved = flop / (score_avg_ms * 1e-3) / 1e12 if score_avg_ms else None
    executed = (scored_pairs * M * 480 / (score_avg_ms * 1e-3) / 1e12) if score_avg_ms and score_variant == 7 else None
    return dict(bound='mfma', achieved=achieved, peak=peak, unit='TFLOP/s',
                frac=(achieved / peak) if achieved else None,
                frac_denominator=('%.1f TFLOP/s = dense f16 MFMA peak 2516.6 / 3 (an f32-class product costs three f16 products): a modelling choice, not a '
                                  'hardware roof; the raw-peak fractions are beside it' % peak) if score_variant == 7 else '%.1f TFLOP/s' % peak,
                frac_of_f16_dense_peak=(achieved / BF16_MFMA_PEAK_TFLOPS) if achieved else None,
                frac_executed=(executed / BF16_MFMA_PEAK_TFLOPS) if executed else None,
                frac_of_f32_mfma_peak=(achieved / FP32_VECTOR_PEAK_TFLOPS) if achieved else None,
                traffic=traffic,
                traffic_source='separate rocprofv3 --pmc passes of this command (FETCH_SIZE; WRITE_SIZE), per launch of the scoring kernel, '
                               'corrected as MI355X_MICROARCH.md prescribes for gfx950: 2 x FETCH_SIZE (the parameter stream is 16-B-per-lane '
                               'global_load_lds, tallied at half its bytes) + WRITE_SIZE; raw counters in traffic_raw; summary in profiles/',
                traffic_raw=traffic_raw,
                traffic_over_algorithmic=(traffic / alg_bytes) if traffic else None,
                kernel=name, kernel_avg_ms=score_avg_ms, launches=launches,
                flop_per_launch=flop, executed_mfma_tflops=executed,
                scored_pairs=scored_pairs, label_pairs=label_pairs,
                note=note + '  SURVEY 8(d) priced this path against the 157.3 TFLOP/s FP32 vector roof; the contraction now runs on the '
                            'f16 matrix pipe, so that roof no longer applies (frac_of_f32_mfma_peak > 1).  frac uses peak = f16 dense MFMA peak / 3; '
                            'frac_of_f16_dense_peak and frac_executed use the raw 2516.6.',
                hbm_algorithmic_bytes_per_launch=alg_bytes,
                hbm_frac=(alg_bytes / (score_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if score_avg_ms else None,
                fb_kernel_avg_ms=fb_avg_ms, fb_kernel_alone_ms=dp_alone_ms,
                fb_note='fb_kernel_avg_ms = HIP-event span of the forward-backward launches INSIDE the timed loop, where their workgroups wait for '
                        'register-file room beside the next step\'s scoring waves (the step time does not wait for them); fb_kernel_alone_ms = the same '
                        'kernels with nothing beside them')


# ------------------------------------------------------------------------------------------------
def bench_decode(args, rank, world, local):
    """--workload C5shard: BASELINE config 5 per GPU -- 417 utterances x 300 frames (1/8 of the 1M-frame corpus), every one of the
    549 GMM states x 4096 mixtures scored for every frame, then frame-synchronous token passing over a synthetic 20 k-word
    pronunciation tree (the reference ships no word list), at most --max-tokens live tokens per utterance.  A step = score +
    decode + results on the host for one resident shard; the ranks work on their own shards with no collective (weak scaling)."""
    from poccala_amd import Engine, PCL_F32, synth
    from poccala_amd.distributed import Control
    from poccala_amd.engine import device_count
    c = dict(synth.CONFIGS['C5shard'])
    if args.utts:
        c['U'] = args.utts
    U, T, D, M, units_n = c['U'], c['T'], c['D'], c['M'], c['units']
    t_setup = time.perf_counter()
    tree, lx = synth.make_pronunciation_tree(args.words, units_n)
    mean, var, w, trans = synth.make_model(units_n, M, D)
    frames, lens, begin = synth.make_frames(U, T, D, seed=1000 * rank)
    # CPU baseline (rank 0, N = 1): the float64 restatement of the same two halves on ONE host core, bounded -- all 549 states x M
    # mixtures of the first frames of one utterance (vectorised NumPy), then the pure-Python token passing over the same tree
    cpu = None
    if args.cpu_baseline and rank == 0 and world == 1:
        from oracle import decoder_oracle as dco
        from oracle import poccala_oracle as po
        nfr = 12
        x = frames[begin[0]:begin[0] + nfr].astype(np.float64)
        t1 = time.perf_counter()
        b_all = np.stack([po.gmm_point(x, mean[j], var[j], w[j]) for j in range(units_n * 3)])
        t_sc = time.perf_counter() - t1
        ndec, ntr = 48, []
        t1 = time.perf_counter()                   # (the decoder's cost does not depend on the values: the scored frames repeat)
        dco.decode(tree, list(trans), np.tile(b_all, (1, ndec // nfr)), beam=0.85, candidate=5, max_tokens=args.max_tokens, trace=ntr)
        t_de = time.perf_counter() - t1
        cpu = dict(value=1.0 / (t_sc / nfr + t_de / ndec), unit='frames/s', cores=1, kind='port',
                   sample='one core: vectorised float64 scoring of all %d states x %d mixtures for the first %d frames of one utterance (%.1f s) + the '
                          'pure-Python restatement of the token passing over the same tree for %d frames (%.1f s, %d live tokens at the end); '
                          'value = 1 / (scoring s per frame + decoding s per frame)' % (units_n * 3, M, nfr, t_sc, ndec, t_de, ntr[-1]))
    ndev = device_count()
    dev = int(os.environ.get('POCCALA_DEVICE', local))
    if bool(os.environ.get('POCCALA_SHARE_DEVICE')) and 0 < ndev < world:
        dev = dev % ndev
    eng = Engine(dev)
    eng.enable_timing(True)
    ctl = Control(rank, world)
    eng.load_model(mean, var, w)
    eng.load_units(np.stack(trans))
    eng.load_lexicon(tree)
    eng.load_frames(frames)
    b = eng.all_state_batch(lens, begin)
    t_setup = time.perf_counter() - t_setup

    def step():
        b.score(PCL_F32)
        return b.decode(max_tokens=args.max_tokens)
    step()
    for _ in range(args.warmup):
        step()
    eng.sync()
    eng.kernel_time('score'); eng.kernel_time('decode')
    ctl.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    eng.sync()
    ctl.barrier()
    elapsed = ctl.allreduce_max(time.perf_counter() - t0)
    sc_ms, k1 = eng.kernel_time('score')
    de_ms, k2 = eng.kernel_time('decode')
    sc_ms, de_ms = sc_ms / max(k1, 1), de_ms / max(k2, 1)
    pairs = U * T * units_n * 3
    flop = pairs * M * (3 * D + 4)
    ntok = np.concatenate([r['n_tokens'] for r in res])
    if rank == 0:
        info = eng.device_info()
        print(json.dumps({
            'metric': 'frames/sec all-state GMM-score + lexicon token-passing decode, 39-d MFCC, 4096-mix', 'value': U * T * world * args.steps / elapsed,
            'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE_NAME, 'data': 'synthetic',
            'config': {'workload': 'C5shard: %d utterances/GPU x %d frames, D=%d, all %d GMM states x %d mixtures for every frame; token passing over a '
                                   'synthetic tree of %d words (%d nodes, %d first-character nodes), <= %d live tokens per utterance, beam 0.85 (Decoder.py:34)'
                                   % (U, T, D, units_n * 3, M, lx.size, len(tree['names']), len(tree['roots']), args.max_tokens),
                       'device': info['name'], 'cus': info['cus'], 'setup_s': t_setup},
            'roofline': dict(bound='mfma', kernel='gmm_score_split16_kernel<39,2>', achieved=flop / (sc_ms * 1e-3) / 1e12, peak=BF16_MFMA_PEAK_TFLOPS / 3, unit='TFLOP/s',
                             frac=flop / (sc_ms * 1e-3) / 1e12 / (BF16_MFMA_PEAK_TFLOPS / 3), kernel_avg_ms=sc_ms, traffic=None,
                             note='the scoring half; the decode half is latency / bandwidth bound: see decode'),
            'decode': dict(kernel='hmm_decode_kernel', kernel_avg_ms=de_ms, token_steps_per_s=float(ntok.sum()) / (de_ms * 1e-3), live_tokens_mean=float(ntok.mean()),
                           live_tokens_max=int(ntok.max()), utterances_at_the_cap=int(sum(r['overflow'] for r in res)),
                           approx_bytes_per_token_step=292, approx_gb_per_s=float(ntok.sum()) * 292 / (de_ms * 1e-3) / 1e9,
                           parity='bit-exact against oracle/decoder_oracle.py (tests/test_gpu_decode.py, incl. this shape).  The reference\'s Decoder.py is dead code: '
                                  'its Token.viterbi recursion, pruning rule, token_passing loop and in-word hand-over are pinned by golden G14 (produced by '
                                  'running those pieces of the reference); first-word seeding, word-to-word hand-over, node-keyed tokens, the finished test '
                                  'on the last emitting state and the order of steps and hand-overs in a frame (rules D1-D5) are the builder\'s '
                                  'completion of that dead code, not the reference\'s semantics'),
            'cpu_baseline': cpu}))
        sys.stdout.flush()
    ctl.barrier()
    b.close()
    ctl.close()
    eng.close()


# ------------------------------------------------------------------------------------------------
def timed_steps(eng, batches, P, align, warmup, steps, barrier):
    """The timed region: W untimed warm-up steps, then exactly K steps between barrier + device sync on both sides.  A step =
    scoring of one resident batch (main stream) + its forward-backward pass loop (second stream, beside the next step's
    scoring); successive steps alternate between the resident batches.  Returns (elapsed s, score kernel ms, launches,
    forward-backward kernel ms, launches) -- the kernel times from HIP events on the library's streams."""
    nb = len(batches)
    step_no = [0]

    def step():
        bt = batches[step_no[0] % nb]
        step_no[0] += 1
        bt.score(P)                                   # main stream
        if align:
            bt.viterbi()                              # BASELINE config 3: forced alignment instead of the Baum-Welch pass
        else:
            bt.forward_backward(fix_pi=False)         # second stream: runs beside the next step's scoring

    # setup, not a step: every resident batch once, so that lazy allocations (tile lists, alpha/beta/xi buffers) never land
    # in a timed step whatever --warmup is
    for bt in batches:
        bt.score(P)
        if align:
            bt.viterbi()
        else:
            bt.forward_backward(fix_pi=False)
    eng.sync()
    for _ in range(warmup):
        step()
    eng.sync()
    eng.kernel_time('score')
    eng.kernel_time('fb')
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    eng.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    score_ms, score_n = eng.kernel_time('score')
    fb_ms, fb_n = eng.kernel_time('fb')
    return elapsed, score_ms, score_n, fb_ms, fb_n



XGMI_LINK_GBS, XGMI_LINKS = 153.0, 7          # MI355X_MICROARCH.md: 7 point-to-point links per GPU, ~153 GB/s each way


def exchange_wire_model(cfg, n, payload_bytes):
    """What one E-step exchange puts on the wire per rank at world size n, and what xGMI would need for it -- a PREDICTION to hold
    the first real multi-GPU line against (no N > 1 timing exists from this box).  Statistics: J M (2D + 1) + J values,
    reduce-scattered by state range (each rank sends and receives (n-1)/n of the block); model: J M (2D + 1) values,
    all-gathered (each rank receives (n-1)/n).  Bounds: every link busy at once (a direct exchange: each of the n-1 peers' shards on
    its own link) and one link at a time (a single ring)."""
    J, M, D = cfg['units'] * 3, cfg['M'], cfg['D']
    stats = (J * M * (2 * D + 1) + J) * payload_bytes
    model = J * M * (2 * D + 1) * payload_bytes
    f = (n - 1) / n if n > 1 else 0.0
    rs, ag = stats * f, model * f
    links = min(XGMI_LINKS, max(1, n - 1))
    return dict(world=n, payload_bytes_per_value=payload_bytes, reduce_scatter_bytes_per_rank=rs, all_gather_bytes_per_rank=ag,
                predicted_ms_all_links=(rs + ag) / (links * XGMI_LINK_GBS * 1e9) * 1e3 if n > 1 else 0.0,
                predicted_ms_one_link=(rs + ag) / (XGMI_LINK_GBS * 1e9) * 1e3 if n > 1 else 0.0,
                note='prediction from the guide\'s link figures, not a measurement; + the per-unit transition accumulators (%d doubles, two all-reduces)' % (cfg['units'] * 18))


def sustained_loop(eng, batches, P, seconds, block=100):
    """The headline's loop (score + forward-backward, the resident batches in alternation) for at least `seconds`: blocks of
    `block` steps, a device sync after each; while a block is queued the shader clock the chip actually holds is probed on the
    device (Engine.clock_probe: s_memtime against the 100 MHz s_memrealtime, 2 ms beside the scoring kernel)."""
    nb = len(batches)
    n = [0]

    def step():
        bt = batches[n[0] % nb]
        n[0] += 1
        bt.score(P)
        bt.forward_backward(fix_pi=False)
    blocks, clocks = [], []
    eng.sync()
    t_all = time.perf_counter()
    while time.perf_counter() - t_all < seconds:
        t0 = time.perf_counter()
        for _ in range(block):
            step()
        try:
            clocks.append(eng.clock_probe(2000))
        except Exception:                      # noqa: a probe that fails must not cost the figure
            pass
        eng.sync()
        blocks.append(time.perf_counter() - t0)
    total = time.perf_counter() - t_all
    return n[0], total, blocks, clocks


def fresh_batch_loop(eng, P, cfg, lens_all, begin_all, steps, rank=0, warm=8, depth=3, label_sets=8, resident=None, rounds=2):
    """The headline's loop as a corpus sweep runs it: the reference hands every worker a NEW (label, data) (AcousticModel.py:664-681
    generator, :861-870 fan-out), so every step here CREATES its label batch (pcl_batch_create_labels: the sentence HMMs, the
    state-major work lists, the scoring tiles at the first score, every lazily allocated buffer) from labels the library has not
    seen in that form, scores it, runs its forward-backward, queues its ln P(O) for the host and -- once the results of the batch
    of `depth` steps ago have landed -- DROPS that batch, all inside the timed region, while the GPU works on the previous steps.
    Frames stay resident (the contract's `value`; the PCIe-inclusive loop moves them too).
    `resident` (the headline's batches): the same number of steps on THEM right before each fresh block (`rounds` blocks of each,
    alternating), so that the ratio compares two loops in the same thermal / clock state -- after the 10-second sustained loop the
    chip runs ~3 % below the short headline loop.  Returns a dict; `value` is frames/s of this rank."""
    from poccala_amd import synth
    U, T = cfg['U'], cfg['T']
    nb = len(lens_all) // U
    sets = [np.stack(synth.make_labels(U, cfg['L'], cfg['units'], seed=31 + 7919 * rank + k)).astype(np.int32) for k in range(label_sets)]
    live, seen = [], [0.0]
    host = dict(create=0.0, enqueue=0.0, close=0.0)

    def one(k, timed):
        lo = U * (k % nb)
        t0 = time.perf_counter()
        b = eng.label_batch(sets[k % label_sets].copy(), lens_all[lo:lo + U], begin_all[lo:lo + U])      # (.copy(): a new array every step)
        t1 = time.perf_counter()
        b.score(P)
        b.forward_backward(fix_pi=False)
        res = b.result_buffers(('logp',), slot=k % (depth + 1))      # the sweep's consumer reads every batch's ln P(O): on its way behind the kernels
        b.fetch_async(res)
        t2 = time.perf_counter()
        live.append((b, res))
        if len(live) > depth:
            old, r = live.pop(0)
            old.fetch_wait()                   # that batch's results are on the host: its work is done, it is dropped and its memory reused
            seen[0] += float(r['logp'][0])
            old.close()
        t3 = time.perf_counter()
        if timed:
            host['create'] += t1 - t0
            host['enqueue'] += t2 - t1
            host['close'] += t3 - t2

    def resident_block(n):
        for k in range(n):
            bt = resident[k % len(resident)]
            bt.score(P)
            bt.forward_backward(fix_pi=False)
    for k in range(warm):
        one(k, False)
    eng.sync()
    t_fresh = t_res = 0.0
    sc_ms = sc_n = 0
    k = warm
    per = max(1, steps // rounds)
    for _ in range(rounds):
        if resident:
            eng.sync()
            t0 = time.perf_counter()
            resident_block(per)
            eng.sync()
            t_res += time.perf_counter() - t0
        eng.kernel_time('score')
        t0 = time.perf_counter()
        for _k in range(per):
            one(k, True)
            k += 1
        eng.sync()
        t_fresh += time.perf_counter() - t0
        a, bn = eng.kernel_time('score')
        sc_ms, sc_n = sc_ms + a, sc_n + bn
    steps = per * rounds
    # the results of a batch made inside the loop against a resident batch of the same labels and frames: the same bits
    last_k = k - 1
    got = live[-1][0].get('logp')
    lo = U * (last_k % nb)
    ref_b = eng.label_batch(sets[last_k % label_sets], lens_all[lo:lo + U], begin_all[lo:lo + U])
    ref_b.score(P); ref_b.forward_backward(fix_pi=False)
    same = bool(np.array_equal(got, ref_b.get('logp')))
    ref_b.close()
    for b, _ in live:
        b.close()
    nfr = int(np.sum(lens_all[:U]))
    out = dict(value=nfr * steps / t_fresh, ms_per_step=t_fresh / steps * 1e3, steps=steps, batches_alive=depth, label_sets=label_sets,
               batch_create_ms=host['create'] / steps * 1e3, enqueue_ms=host['enqueue'] / steps * 1e3, batch_close_ms=host['close'] / steps * 1e3,
               score_kernel_ms=sc_ms / max(sc_n, 1), same_bits_as_a_resident_batch=same,
               what='every step creates its label batch (new labels), scores it, runs forward-backward, queues its ln P(O) for the host, and -- once the '
                    'results of the batch of %d steps ago have landed -- drops that batch, all inside the timed region; batch_create_ms / enqueue_ms / '
                    'batch_close_ms = HOST time per step (the GPU works on the previous steps meanwhile; close includes the wait for the old batch\'s '
                    'results, which is what keeps the host from running ahead of the GPU without bound); frames resident' % depth)
    if resident:
        out['resident_ms_per_step_beside'] = t_res / steps * 1e3
        out['fresh_over_resident'] = t_res / t_fresh
        out['what'] += '; resident_ms_per_step_beside = the headline\'s loop on its resident batches in blocks of %d steps alternating with the fresh blocks ' \
                       '(same clock state); fresh_over_resident = the ratio of the two rates' % per
    return out


def pcie_inclusive_loop(eng, P, cfg, frames, labels_all, lens_all, steps, warm=4):
    """SURVEY 8(d)'s end-to-end protocol on the headline's work, as a corpus sweep: every step's frames come up over PCIe, its label
    batch is CREATED in the step (new labels: AcousticModel.py:664-681 hands every worker a new (label, data)), and its results go
    down, all beside the kernels of the neighbouring steps.  Per step: H2D of the NEXT batch's frames into the frame slot that is
    not being scored (copy stream) | pcl_batch_create_labels | score -> Viterbi (main stream) -> forward-backward (second stream) |
    D2H of ln P(O), ln gamma_t(j) (all of it: 152 MB), the stored ln xi, the Viterbi paths and scores into page-locked buffers
    (download stream) | the batch of three steps ago dropped.  THREE batches and result sets in flight (two frame slots): the
    forward-backward of step k runs beside the scoring of step k+1 and its results leave during step k+2."""
    from poccala_amd import synth
    U, T, NB = cfg['U'], cfg['T'], 3
    nfr = U * T
    legs = set(os.environ.get('POCCALA_PCIE_LEGS', 'h2d,d2h,vit,fresh').split(','))     # diagnosis: drop a leg to see what it costs
    pin = [eng.pinned_empty((nfr, cfg['D']), np.float32) for _ in range(2)]
    for k in range(2):
        pin[k][:] = frames[k * nfr:(k + 1) * nfr]
    begin = np.arange(U, dtype=np.int64) * T
    lens = np.ascontiguousarray(lens_all[:U])
    sets = [np.stack(synth.make_labels(U, cfg['L'], cfg['units'], seed=977 + k)).astype(np.int32) for k in range(8)]
    eng.stage_frames(pin[0])
    eng.swap_frames()
    total = warm + steps
    live = [None] * NB
    res = [None] * NB
    resident = None
    if 'fresh' not in legs:                    # (diagnosis: the round-4 protocol, three resident batches in rotation)
        resident = [eng.label_batch(sets[k], lens, begin) for k in range(NB)]
    eng.stage_frames(pin[0])
    t0 = None
    host_create = 0.0
    for k in range(total):
        if k == warm:
            eng.sync()
            t0 = time.perf_counter()
            host_create = 0.0
            eng.kernel_time('score'); eng.kernel_time('fb')
        eng.swap_frames()                      # chunk k is the current frame matrix (its copy had a whole step to arrive)
        if k + 1 < total and 'h2d' in legs:
            eng.stage_frames(pin[(k + 1) % 2])
        elif k + 1 < total:
            eng.stage_frames(pin[(k + 1) % 2][:1])                              # (diagnosis: a one-row chunk keeps the protocol, moves nothing)
        old = live[k % NB]
        if old is not None and 'd2h' in legs:
            old.fetch_wait()                   # the host is done with that batch's results (a consumer would have read them)
        tc = time.perf_counter()
        if resident is not None:
            bt = resident[k % NB]
        else:
            if old is not None:
                old.close()
            bt = eng.label_batch(sets[k % len(sets)].copy(), lens, begin)
        host_create += time.perf_counter() - tc
        live[k % NB] = bt
        bt.score(P)
        if 'vit' in legs:
            bt.viterbi()
        bt.forward_backward(fix_pi=False)
        if 'd2h' in legs:
            if res[k % NB] is None:            # (page-locked, the engine's: one set per slot in flight)
                res[k % NB] = bt.result_buffers(slot=k % NB)
            bt.fetch_async(res[k % NB] if 'vit' in legs else {q: v for q, v in res[k % NB].items() if q not in ('path', 'point')})
    if 'd2h' in legs:
        for b in live:
            if b is not None:
                b.fetch_wait()
    eng.sync()
    elapsed = time.perf_counter() - t0
    sc_ms, sc_n = eng.kernel_time('score')
    fb_ms, fb_n = eng.kernel_time('fb')
    last = live[(total - 1) % NB]
    ok = None
    if 'd2h' in legs:
        ok = bool(np.array_equal(res[(total - 1) % NB]['logp'], last.get('logp')))
        if 'vit' in legs:
            ok = ok and bool(np.array_equal(res[(total - 1) % NB]['path'], np.concatenate(last.get('path'))))
    bytes_down = int(sum(v.nbytes for v in res[0].values())) if res[0] is not None else 0
    for b in set(x for x in live if x is not None) | set(resident or []):
        b.close()
    eng.load_frames(frames)                    # back to the resident matrix the other measurements index
    return dict(value=nfr * steps / elapsed, ms_per_step=elapsed / steps * 1e3, steps=steps, legs=sorted(legs), h2d_bytes_per_step=int(pin[0].nbytes),
                d2h_bytes_per_step=bytes_down, results_intact=ok, score_kernel_ms=sc_ms / max(sc_n, 1), fb_span_ms=fb_ms / max(fb_n, 1),
                batch_create_ms=host_create / steps * 1e3,
                what='per step: frames H2D (copy stream) | the step\'s label batch created (new labels) | score + Viterbi + forward-backward | ln P(O), '
                     'ln gamma_t(j), stored ln xi, Viterbi paths and scores D2H into page-locked memory (download stream) | the batch of three steps ago '
                     'dropped; wall clock over the whole pipeline, the copies overlapped behind the neighbouring steps\' kernels; three batches / result '
                     'sets in flight')


def engine_with_variant(device, variant):
    """A fresh context whose scoring kernel is PCL_SCORE_VARIANT=variant (read by pcl_init)."""
    from poccala_amd import Engine
    old = os.environ.get('PCL_SCORE_VARIANT')
    os.environ['PCL_SCORE_VARIANT'] = str(variant)
    try:
        return Engine(device)
    finally:
        if old is None:
            os.environ.pop('PCL_SCORE_VARIANT', None)
        else:
            os.environ['PCL_SCORE_VARIANT'] = old


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    args.gpus = world
    if os.environ.get('POCCALA_HANG_DUMP'):       # diagnostics: where is every thread after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ['POCCALA_HANG_DUMP']), repeat=True, file=sys.stderr)

    from poccala_amd import synth
    if args.workload == 'C5shard':                # BASELINE config 5: all-state scoring + lexicon token-passing decode
        return bench_decode(args, rank, world, local)
    if args.workload == 'C4':                     # config 4 at its stated size: 8192 utterances, one whole EM iteration
        return bench_c4_full(args, rank, world, local)
    if args.workload == 'C5':                     # config 5 at its stated size: the 1M-frame corpus streamed
        return bench_c5_full(args, rank, world, local)
    P_name = args.precision
    cfg = dict(synth.CONFIGS[args.workload])
    if args.utts:
        cfg['U'] = args.utts
    if args.mix:
        cfg['M'] = args.mix
    if args.units:
        cfg['units'] = args.units
    reduced = bool(args.utts or args.mix or args.units)
    t_setup = time.perf_counter()
    tl = {}                                           # where the wall-clock of this run goes (extra.timeline_s)
    t_mark = time.perf_counter()
    mean, var, w, trans = synth.make_model(cfg['units'], cfg['M'], cfg['D'], seed=1)
    nb = max(1, args.batches)
    frames, lens_all, begin_all = synth.make_frames(cfg['U'] * nb, cfg['T'], cfg['D'], seed=1000 * rank)      # each rank its own shard
    labels_all = synth.make_labels(cfg['U'] * nb, cfg['L'], cfg['units'], seed=2 + 7919 * rank)
    lens, begin, labels = lens_all[:cfg['U']], begin_all[:cfg['U']], labels_all[:cfg['U']]      # batch 0: accounting and the CPU leg
    tl['synthetic_model_and_frames_s'] = time.perf_counter() - t_mark
    t_mark = time.perf_counter()

    # CPU baseline leg first: its worker pool is forked BEFORE this process initialises HIP
    cpu = None
    if args.cpu_baseline and rank == 0 and world == 1:
        cpu = cpu_baseline(cfg, mean, var, w, trans, frames, lens, begin, labels)
    tl['cpu_baseline_s'] = time.perf_counter() - t_mark
    t_mark = time.perf_counter()

    from poccala_amd import Engine, PCL_F32, PCL_F64
    from poccala_amd.engine import device_count
    P = PCL_F32 if P_name == 'f32' else PCL_F64
    ndev = device_count()
    dev = int(os.environ.get('POCCALA_DEVICE', local))
    # rehearsal of the N > 1 path on a box with fewer GPUs than ranks (POCCALA_SHARE_DEVICE=1): several ranks per device, and
    # on EVERY rank the host transport in place of RCCL (which refuses two ranks on one device)
    shared = bool(os.environ.get('POCCALA_SHARE_DEVICE')) and 0 < ndev < world
    if shared:
        dev = dev % ndev
    elif dev >= ndev:
        sys.exit('bench.py: rank %d needs HIP device %d but %d are visible (POCCALA_SHARE_DEVICE=1 rehearses several ranks per device)' % (rank, dev, ndev))
    eng = Engine(dev)
    eng.enable_timing(True)
    # control plane: a few tiny host-side exchanges over TCP (poccala_amd.distributed.Control), no torch in
    # the GPU processes; the statistics themselves travel over RCCL inside the library.
    from poccala_amd.distributed import Control
    ctl = Control(rank, world)
    barrier = ctl.barrier

    eng.load_model(mean, var, w)
    eng.load_units(np.stack(trans))
    eng.load_frames(frames)
    batches = []
    for k in range(nb):
        lo, hi = cfg['U'] * k, cfg['U'] * (k + 1)
        batches.append(eng.label_batch(labels_all[lo:hi], lens_all[lo:hi], begin_all[lo:hi]))     # AcousticModel.embedded, in the library
    n_states = batches[0].N
    # (the communicator is made AFTER the timed loop, under the watchdog below: the headline loop has no collective in it, and an RCCL
    #  bootstrap that fails or never returns on some node must not take the line with it)
    comm = eng.comm_info()
    t_setup = time.perf_counter() - t_setup
    tl['upload_and_batches_s'] = time.perf_counter() - t_mark
    t_mark = time.perf_counter()

    align = args.workload == 'C3'                     # score + Viterbi forced alignment (the task BASELINE config 3 names)
    if align:
        args.extra = 0                                # (the extras measure the E-step)
    elapsed, score_ms, score_n, fb_ms, fb_n = timed_steps(eng, batches, P, align, args.warmup, args.steps, barrier)
    elapsed = ctl.allreduce_max(elapsed)
    if os.environ.get('POCCALA_TEST_DIE_RANK') == str(rank):      # tests/test_gpu_a_bench_ranks.py: a rank that dies behind the timed loop
        os._exit(7)
    # the dynamic-programming kernels ALONE (outside the timed region): inside the loop their HIP events span the time their
    # workgroups wait for register-file room beside the next step's scoring waves, not the time they work
    dp_alone_ms = None
    try:
        for _ in range(3):
            (batches[0].viterbi if align else batches[0].forward_backward)()
        eng.sync()
        eng.kernel_time('viterbi' if align else 'fb')
        for _ in range(5):
            (batches[0].viterbi if align else batches[0].forward_backward)()
        eng.sync()
        ms_, n_ = eng.kernel_time('viterbi' if align else 'fb')
        dp_alone_ms = ms_ / max(n_, 1)
    except Exception:                          # noqa: never the headline's problem
        pass
    tl['timed_loop_s'] = time.perf_counter() - t_mark
    t_mark = time.perf_counter()

    frames_per_rank = int(lens.sum())
    total_frames = frames_per_rank * world
    value = total_frames * args.steps / elapsed

    # dominant kernel: gmm_score.  Algorithmic FLOP per launch = scored (frame, state) pairs x M x (3D+4)
    # (SURVEY.md section 8d); the emitting rows of an utterance are N-2.
    score_variant = int(os.environ.get('PCL_SCORE_VARIANT', '7'))
    if P == PCL_F64:
        score_variant = 0
    score_kernel_name, score_peak, score_note = KERNELS.get(score_variant, KERNELS[1])
    pairs = int(((n_states - 2).astype(np.int64) * lens.astype(np.int64)).sum())
    # a label that names a unit twice has the same (frames, state) pair on two rows: the library scores it ONCE and copies the row
    # (the reference scores it once per label position).  The scoring kernel's rate is quoted on what it computed.
    # (mean over the resident batches: the kernel's average launch time is over all of them)
    scored_pairs = int(round(sum(3 * len(set(np.asarray(lab).tolist())) * int(t) for lab, t in zip(labels_all, lens_all)) / nb))
    score_avg_ms = score_ms / max(score_n, 1)
    # algorithmic HBM bytes per scoring launch: frames once + parameters of the states with work once + B written once
    alg_bytes = frames_per_rank * cfg['D'] * 4 + len(set(np.concatenate(labels).tolist())) * 3 * cfg['M'] * (2 * cfg['D'] + 1) * 4 + scored_pairs * 8
    traffic, traffic_raw = args.traffic_bytes, None
    if traffic is None and args.workload == 'C4shard' and not reduced and P == PCL_F32 and score_variant == 7:
        traffic, traffic_raw = committed_traffic()
    roofline = make_roofline(score_variant, score_avg_ms if score_n else None, score_n, scored_pairs, pairs, cfg['M'], cfg['D'], alg_bytes, traffic, traffic_raw,
                             fb_ms / max(fb_n, 1), dp_alone_ms)

    info = eng.device_info()                   # (not from the watchdog thread: the main thread may be inside the runtime)

    # beside the headline, from the same resident batches (outside the timed region): the same loop held for >= 10 s, and the
    # end-to-end pipeline of SURVEY 8(d) with every step's frames and results crossing PCIe
    sustained = pcie = fresh = None
    loops = {}                                 # sustained / fresh / pcie as they complete (make_full reads them)
    def make_full(extra):
        out = {
            'metric': 'frames/sec GMM-score+Viterbi forced alignment, 39-d MFCC, 2048-mix' if align else 'frames/sec GMM-score+fwd-bwd, 39-d MFCC, 2048-mix',
            'value': value, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': DTYPE_NAME if P == PCL_F32 else 'f64', 'data': 'synthetic',
            'config': {'workload': ('REDUCED for a rehearsal, not the named configuration -- ' if reduced else '') + ('%s: %d utterances/GPU x %d frames, D=%d MFCC, M=%d mixtures, %d units (J=%d tied GMM states), '
                                    'L=%d units/utterance (N=%d-state sentence HMMs); GMM scoring of the %d label states of every frame '
                                    + ('+ Viterbi forced alignment (LHMM.viterbi, bit-exact given the emissions)' if align else
                                       '+ Baum-Welch forward-backward pass loop (3 passes, xi/gamma/pi/posteriors)'))
                                   % (args.workload, cfg['U'], cfg['T'], cfg['D'], cfg['M'], cfg['units'], cfg['units'] * 3, cfg['L'],
                                      3 * cfg['L'] + 2, 3 * cfg['L']),
                       'utterances_total': cfg['U'] * world, 'frames_per_step_total': total_frames,
                       'resident_batches': nb,
                       'pipeline': ('forward-backward of step k on a second HIP stream beside the scoring of step k+1 (steps alternate '
                                    'between %d resident batches)' % nb) if nb > 1 else 'score and forward-backward of a step serialise (one resident batch)',
                       'arithmetic': ('f32-CLASS Gaussian scoring: every operand as two f16 pieces (22 significand bits), three f16 MFMA products per '
                                      'f32 product, f32 accumulate (measured |d ln b| vs float64 1.5e-5 at |ln b| ~ 85; the exact f32 chain gives '
                                      '1.0e-5); f64 dynamic programming.  The strict-f32 kernel (v_mfma_f32_32x32x2_f32) is timed under extra.strict_f32'
                                      ) if P == PCL_F32 else 'f64',
                       'transport': comm['transport'], 'rccl_nranks': comm['rccl_nranks'],
                       'device': info['name'], 'cus': info['cus']},
            'roofline': roofline,
            'cpu_baseline': cpu,
        }
        sustained, fresh, pcie = loops.get('sustained'), loops.get('fresh'), loops.get('pcie')
        rf = out['roofline'] = dict(roofline)      # the loop-level figures the driver's record should keep (it keeps this object whole)
        rf['sustained_value'] = sustained.get('value') if sustained else None
        rf['fresh_batches_value'] = fresh.get('value') if fresh else None
        rf['pcie_inclusive_value'] = pcie.get('value') if pcie else None
        sf_ = (extra or {}).get('strict_f32') or {}
        rf['strict_f32_value'], rf['strict_f32_frac'] = sf_.get('value'), sf_.get('frac')
        em_ = (extra or {}).get('em_shaped_models') or {}      # the headline loop on the models EM leaves (VERDICT r5 next #2b)
        for key in ('value_em2_model', 'value_em3_model', 'off_pipe_mixture_share_em2', 'off_pipe_mixture_share_em3'):
            rf[key] = em_.get(key)
        if sustained:
            out['value_sustained'] = sustained.get('value')
            out['sustained'] = sustained
        if fresh:
            out['value_fresh_batches'] = fresh.get('value')
            out['fresh_batches'] = fresh
        if pcie:
            out['value_pcie_inclusive'] = pcie.get('value')
            out['pcie_inclusive'] = pcie
        if extra:
            out['extra'] = extra
            if extra.get('roofline_estep'):
                out['roofline_estep'] = extra.pop('roofline_estep')
            sf = extra.get('strict_f32')
            if sf and sf.get('value'):             # the same timed loop on the strict-f32 kernel (a second engine, PCL_SCORE_VARIANT=3)
                out['value_strict_f32'] = sf['value']
                out['ms_per_step_strict_f32'] = sf['ms_per_step']
        if cpu:
            out['gpu_over_cpu'] = {'vs_cpu_baseline_value': value / cpu['value'], 'vs_blas_gemm_formulation': value / cpu['gemm_value'],
                                   'vs_reference_arithmetic_vectorised': value / cpu['vectorised_value'], 'vs_reference_as_written': value / cpu['faithful_value'],
                                   'note': 'a GPU / CPU ratio says nothing about kernel quality (the roofline fractions do); north_star asks >= 200 against the reference CPU path'}
        return out

    # The headline, its roofline and the CPU leg are final: the line goes out NOW (rank 0), before anything else can hang, fail or
    # outgrow a reader; it is printed again at the end with what the untimed measurements added (the driver reads the last line).
    if rank == 0:
        emit(make_full(None), final=False)
    # What follows (the held loops, the E-step with its exchange, side measurements) is outside the timed region and must not be able
    # to take the line with it: an exception is reported inside `extra`, and if it does not come back within --extra-timeout seconds
    # (a collective that never completes on some node), rank 0 prints the line with what it has and the job exits non-zero.
    printed = threading.Event()

    def give_up():
        if rank == 0 and not printed.is_set():
            emit(make_full(dict(error='extras did not finish within %d s' % args.extra_timeout)), final=False)
        os._exit(3)                            # the line is out, but the job did NOT complete: launchers must see a failure
    dog = threading.Timer(args.extra_timeout, give_up)
    dog.daemon = True
    dog.start()
    if args.sustain > 0 and not align and P == PCL_F32:
        t_mark2 = time.perf_counter()
        try:
            ns, tot, blocks, clocks = sustained_loop(eng, batches, P, args.sustain)
            tot = ctl.allreduce_max(tot)
            loops['sustained'] = sustained = dict(value=frames_per_rank * world * ns / tot, seconds=tot, steps=ns,
                             ms_per_100_steps=dict(min=min(blocks) * 1e3, median=float(np.median(blocks)) * 1e3, max=max(blocks) * 1e3),
                             shader_mhz=dict(min=min(clocks), median=float(np.median(clocks)), max=max(clocks), probes=len(clocks)) if clocks else None,
                             clock_source='on-device probe beside the scoring kernel: s_memtime (shader cycles) / s_memrealtime (100 MHz) over 2 ms, once per '
                                          '100 steps; the PMC figure (GRBM_GUI_ACTIVE per kernel) is in profiles/; rocm-smi sclk is the requested level, not this')
        except Exception as e:                 # noqa: never the headline's problem
            loops['sustained'] = sustained = dict(error=repr(e))
        tl['sustained_loop_s'] = time.perf_counter() - t_mark2
        t_mark2 = time.perf_counter()
        try:
            loops['fresh'] = fresh = fresh_batch_loop(eng, P, cfg, lens_all, begin_all, max(4 * args.steps, 80), rank=rank, resident=batches)
            fresh['value'] = fresh['value'] * world        # (whole job: every rank sweeps its own shard; rank 0's clock)
        except Exception as e:                 # noqa: never the headline's problem
            loops['fresh'] = fresh = dict(error=repr(e))
        tl['fresh_batch_loop_s'] = time.perf_counter() - t_mark2
        t_mark2 = time.perf_counter()
        if world == 1 and not reduced and nb >= 2:
            try:
                loops['pcie'] = pcie = pcie_inclusive_loop(eng, P, cfg, frames, labels_all, lens_all, max(args.steps, 20))
            except Exception as e:             # noqa
                loops['pcie'] = pcie = dict(error=repr(e))
        tl['pcie_inclusive_loop_s'] = time.perf_counter() - t_mark2

    extra = None
    comm_error = None
    if world > 1 or os.environ.get('POCCALA_FORCE_DIST'):          # FORCE_DIST: exercise RCCL at world 1
        if os.environ.get('POCCALA_TEST_HANG_COMM'):               # tests: a communicator bootstrap that never returns -> the watchdog's job
            time.sleep(1e6)
        try:
            if shared or os.environ.get('POCCALA_NO_RCCL'):
                eng.comm_init_host(rank, world, ctl.allgather_bytes)
            else:
                eng.comm_init(rank, world, ctl.broadcast(eng.comm_unique_id() if rank == 0 else None, src=0))
            comm = eng.comm_info()
        except Exception as e:                 # noqa: reported, the E-step extras (which need the exchange) are skipped
            import traceback
            traceback.print_exc()
            comm_error = 'communicator: %s: %s' % (type(e).__name__, e)
    if comm_error:
        extra = dict(error=comm_error)
    elif args.extra:
        try:
            extra = extras(args, eng, ctl, batches, labels, frames, frames_per_rank, total_frames, elapsed, P, cfg, pairs, t_setup, mean, var, w, trans,
                           dict(labels_all=labels_all, lens_all=lens_all, begin_all=begin_all, tl=tl, t_mark=t_mark))
        except Exception as e:                 # noqa: the headline does not depend on the extras
            import traceback
            traceback.print_exc()
            extra = dict(error='%s: %s' % (type(e).__name__, e))
    if rank == 0:
        emit(make_full(extra), final=True)
        printed.set()
    barrier()
    for bt in batches:
        bt.close()
    eng._lib.pcl_comm_destroy(eng._ctx)
    ctl.close()
    eng.close()
    dog.cancel()


def scoring_kernel_sha16():
    """Identity of the dominant kernel's CODE: sha256 over the anonymous namespace of gmm_score_split.hip -- the kernel, its helpers and
    its compile-time constants -- with comments and white space removed, plus the compiler flags the Makefile gives that file.  The
    launch side of the file (tile size helpers, the host launcher) is not part of it: an edit there cannot change what a PMC pass
    measured (round 4 hashed the whole file and a launch-side commit nulled the traffic figure of the driver's line)."""
    import hashlib
    import re
    src = open(os.path.join(ROOT, 'poccala_amd', 'csrc', 'gmm_score_split.hip')).read()
    body = src[src.index('namespace {'):src.index('}  // namespace')]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    body = re.sub(r'//[^\n]*', '', body)
    body = re.sub(r'\s+', ' ', body).strip()
    mk = open(os.path.join(ROOT, 'poccala_amd', 'csrc', 'Makefile')).read()
    flags = ' '.join(sorted(l.split('=', 1)[1].strip() for l in mk.splitlines() if l.startswith('CXXFLAGS =') or l.startswith('FLAGS_gmm_score_split')))
    return hashlib.sha256((body + '\x00' + flags).encode()).hexdigest()[:16]


def committed_traffic():
    """HBM bytes per scoring launch from the committed PMC passes of this command (counters cannot be read from inside the
    run): (corrected bytes, {'FETCH_SIZE_bytes', 'WRITE_SIZE_bytes', 'file', ...}) or (None, {'stale': ...}).  The summary
    records the identity of the kernel code it was taken from (scoring_kernel_sha16); when the kernel has changed since, the figure
    is withheld (traffic = null) instead of being passed off as a measurement of the current kernel."""
    try:
        now = scoring_kernel_sha16()
    except (OSError, ValueError):
        now = None
    import glob
    names = sorted((os.path.basename(f) for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_summary.txt'))), reverse=True)
    for name in names:
        try:
            fetch = write = sha = None
            for line in open(os.path.join(ROOT, 'profiles', name)):
                if 'gmm_score_split16_kernel' in line and 'FETCH_SIZE' in line:
                    fetch = float(line.split('per-dispatch=')[1]) * 1024.0            # rocprofv3 reports KiB
                if 'gmm_score_split16_kernel' in line and 'WRITE_SIZE' in line:
                    write = float(line.split('per-dispatch=')[1]) * 1024.0
                if line.startswith('kernel_code_sha16'):
                    sha = line.split()[-1]
            if fetch is not None and write is not None:
                if sha is None or now is None or sha != now:
                    return None, dict(stale='profiles/%s was taken from kernel code %s, the kernel is now %s: re-run tools/gpu_profile.sh'
                                            % (name, sha, now))
                return 2.0 * fetch + write, dict(FETCH_SIZE_bytes=fetch, WRITE_SIZE_bytes=write, file='profiles/' + name, kernel_code_sha16=sha)
        except (OSError, ValueError, IndexError):
            pass
    return None, None


def extras(args, eng, ctl, batches, labels, frames, frames_per_rank, total_frames, elapsed, P, cfg, pairs, t_setup, mean, var, w, trans, more):
    """Untimed-region measurements: forced alignment, the full E-step with the statistics exchange and both M-steps, PCIe legs."""
    from poccala_amd import PCL_F32, PCL_F64
    barrier = ctl.barrier
    world = ctl.world
    eng.sync()
    barrier()
    batch = batches[0]
    eng.kernel_time('viterbi')                 # (the PCIe-inclusive loop left its launches in the timer)
    t1 = time.perf_counter()
    batch.viterbi()
    eng.sync()
    t_vit = time.perf_counter() - t1
    # what follows alignment in training scheme 1 (next row f2): per-frame unit and GMM-state assignment
    ids = [np.repeat(np.asarray(lab, dtype=np.int32), 3) for lab in labels]
    row_unit = [np.concatenate([[i[0]], i, [i[-1]]]).astype(np.int32) for i in ids]
    t1 = time.perf_counter()
    batch.regroup(row_unit, 3)
    t_regroup = time.perf_counter() - t1
    rg_ms, _ = eng.kernel_time('regroup')
    vit_ms, vit_n = eng.kernel_time('viterbi')
    vit_ms /= max(vit_n, 1)
    # PCIe legs the timed region excludes (the whole resident frame matrix, per batch)
    t1 = time.perf_counter()
    eng.load_frames(frames)
    t_h2d = (time.perf_counter() - t1) / len(batches)
    # full E-step: score -> forward-backward -> GMM statistics + per-unit transition accumulators -> exchange
    # (reduce-scatter by state range -> M-step on the owned states -> all-gather of the model) -> transition M-step
    payload = resolve_payload(args, world)
    payload_name = 'f32' if payload == PCL_F32 else 'f64'
    batch.accumulate(P)                       # setup, not measured: the accumulate pass's work lists and tile-image buffers are
    batch.accumulate_hmm()                    # allocated on first use
    eng.sync()
    # one untimed E-step right in front of the timed one: the host-side measurements above left the GPU idle for ~100 ms and the
    # first passes after that run below their steady rate while the clocks come back (accumulate 51 ms instead of 42: tools/acc_insitu.py)
    eng.stats_zero()
    batch.score(P); batch.forward_backward(fix_pi=False); batch.accumulate(P); batch.accumulate_hmm()
    eng.sync()
    for k in ('accumulate', 'hmm_acc', 'reduce_scatter', 'mstep_owned', 'all_gather', 'derive', 'score', 'fb'):
        eng.kernel_time(k)
    eng.stats_zero()
    eng.sync()
    barrier()
    t1 = time.perf_counter()
    batch.score(P)
    batch.forward_backward(fix_pi=False)
    batch.accumulate(P)
    batch.accumulate_hmm()
    eng.sync()
    t_local = time.perf_counter() - t1
    eng.em_exchange(1e-3, payload, True)
    eng.sync()
    t_rank = time.perf_counter() - t1
    barrier()
    t_estep = time.perf_counter() - t1
    kt = {k: eng.kernel_time(k)[0] for k in ('accumulate', 'hmm_acc', 'reduce_scatter', 'mstep_owned', 'all_gather', 'derive', 'score', 'fb')}
    per_rank = ctl.allgather(dict(rank=ctl.rank, estep_local_ms=t_local * 1e3, estep_ms=t_rank * 1e3, exchange_ms=(t_rank - t_local) * 1e3,
                                  reduce_scatter_ms=kt['reduce_scatter'], mstep_owned_ms=kt['mstep_owned'], all_gather_ms=kt['all_gather'],
                                  derive_ms=kt['derive'], accumulate_ms=kt['accumulate'], hmm_acc_ms=kt['hmm_acc']))
    # the same E-step from the same model with the exchange PIPELINED behind the accumulate pass (pcl_batch_accumulate_exchange: state
    # chunks leave for reduce-scatter -> M-step -> all-gather -> derive as soon as the pass is done with them)
    pipe = None
    try:
        eng.load_model(mean, var, w)
        eng.load_units(np.stack(trans))
        batch.refresh_transitions()
        eng.stats_zero()
        batch.score(P); batch.forward_backward(fix_pi=False); batch.accumulate(P); batch.accumulate_hmm()     # (clocks, buffers)
        eng.sync()
        for k in ('accumulate', 'reduce_scatter', 'mstep_owned', 'all_gather', 'derive'):
            eng.kernel_time(k)
        eng.stats_zero()
        eng.sync()
        barrier()
        t1 = time.perf_counter()
        batch.score(P)
        batch.forward_backward(fix_pi=False)
        batch.accumulate_hmm()
        os.environ['PCL_PIPE_MODE'] = '0' if world == 1 else '1'   # one GPU has no reduce-scatter to send early: measure the full chain there
        batch.accumulate_exchange(P, 1e-3, payload, True, n_chunks=8)
        os.environ.pop('PCL_PIPE_MODE', None)
        eng.sync()
        t_prank = time.perf_counter() - t1
        barrier()
        t_pipe = time.perf_counter() - t1
        kp = {k: eng.kernel_time(k)[0] for k in ('accumulate', 'reduce_scatter', 'mstep_owned', 'all_gather', 'derive')}
        pipe = dict(estep_ms=t_pipe * 1e3, frames_per_s=total_frames / t_pipe, n_chunks=8,
                    mode='whole chain per chunk (PCL_PIPE_MODE=0)' if world == 1 else 'reduce-scatter per chunk, M-step / all-gather / derive at the end (default)',
                    per_rank=ctl.allgather(dict(rank=ctl.rank, estep_ms=t_prank * 1e3, accumulate_ms=kp['accumulate'], reduce_scatter_ms=kp['reduce_scatter'],
                                                mstep_owned_ms=kp['mstep_owned'], all_gather_ms=kp['all_gather'], derive_ms=kp['derive'])),
                    what='score -> forward-backward -> per-unit merge -> accumulate with the exchange pipelined: the kernel times are sums over the 8 '
                         'chunks and overlap the accumulate pass (accumulate_ms is stretched by what runs beside it); estep_ms against extra.estep_ms '
                         'is what the pipelining buys')
    except Exception as e:                     # noqa: never the headline's problem
        pipe = dict(error=repr(e))
    t1 = time.perf_counter()
    batch.get('logp'); batch.get('gamma'); eng.hmm_acc_download()
    t_d2h = time.perf_counter() - t1
    # how many (frame, state) pairs the exact underflow compaction kept (what the accumulate kernels actually processed)
    lgam = batch.get('lgamma')
    survive = float(np.mean([np.mean(l[1:-1] >= -150 * np.log(2)) for l in lgam[::16]]))
    del lgam
    # E-step accounting: accumulate = recompute + two weighted moments = M (7D + 8) flop per (frame, state) pair
    acc_ms = kt['accumulate']
    acc_flop = pairs * cfg['M'] * (7 * cfg['D'] + 8)
    extra = dict(viterbi_frames_per_s_per_gpu=frames_per_rank / t_vit, viterbi_kernel_ms=vit_ms,
                 regroup_kernel_ms=rg_ms, regroup_call_ms=t_regroup * 1e3,
                 estep_frames_per_s=total_frames / t_estep, estep_ms=t_estep * 1e3, accumulate_ms=acc_ms, hmm_acc_ms=kt['hmm_acc'],
                 exchange=dict(payload=payload_name, per_rank=per_rank,
                               wire=exchange_wire_model(cfg, world, 4 if payload == PCL_F32 else 8),
                               wire_at_8_gpus_f32=exchange_wire_model(cfg, 8, 4),
                               what='reduce-scatter of the GMM statistics by state range -> GMM.update_param on the owned J/N states -> '
                                    'all-gather of (mean, var, weight) -> layouts re-derived; per-unit transition accumulators merged by max + sum '
                                    'all-reduces, transition M-step on every rank; one rank: the M-step alone'),
                 estep_pipelined=pipe,
                 frames_h2d_ms=t_h2d * 1e3, results_d2h_ms=t_d2h * 1e3,
                 setup_s=t_setup,
                 roofline_estep=dict(kernel='acc16_consumer_kernel<39> (+ acc16_producer_kernel<39>, compaction)', ms=acc_ms, bound='mfma',
                                     flop_per_launch=acc_flop, algorithmic_tflops=acc_flop / (acc_ms * 1e-3) / 1e12 if acc_ms else None,
                                     surviving_pair_fraction=survive,
                                     executed_mfma_tflops=(survive * pairs * cfg['M'] / 1024.0 * 33 * 32768 / (acc_ms * 1e-3) / 1e12) if acc_ms else None,
                                     peak=BF16_MFMA_PEAK_TFLOPS, frac_executed=(survive * pairs * cfg['M'] / 1024.0 * 33 * 32768 / (acc_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS) if acc_ms else None,
                                     note='algorithmic = M (7D + 8) flop per (frame, state) pair over ALL pairs (SURVEY 8d); executed = 33 MFMAs of '
                                          '32x32x16 f16 (15 for the recomputed exponent, 18 for the two weighted moments: posteriors and features in two f16 '
                                          'pieces each, per-mixture running power-of-two scale; round 2: 45 with bf16 x3 moments) per 32 frames x 32 mixtures of '
                                          'the SURVIVING pairs.  Bench features are random N(0,1): flat posteriors, ~78 % of the pairs survive the exact '
                                          'underflow compaction; aligned speech is peaked: extra.estep_peaked'))
    if ctl.world == 1 and P == PCL_F32 and not (args.utts or args.mix or args.units):
        more['frames'] = frames
        more['batches'] = batches
        extra.update(side_measurements(args, eng, cfg, labels, mean, var, w, trans, frames_per_rank, pairs, more))
        more['tl']['total_before_line_s'] = time.perf_counter() - T_PROCESS_START
        extra['timeline_s'] = more['tl']
    return extra


def smi_sample():
    """One reading of the chip's clock and power from rocm-smi (None when the tool or the permission is missing)."""
    try:
        out = subprocess.run(['rocm-smi', '-d', '0', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=5).stdout
        card = next(iter(json.loads(out).values()))
        sclk = next((v for k, v in card.items() if 'sclk' in k.lower()), None)
        power = next((v for k, v in card.items() if 'power' in k.lower() and 'socket' in k.lower()), None) or \
            next((v for k, v in card.items() if 'power' in k.lower()), None)
        mhz = float(str(sclk).strip('()').lower().replace('mhz', '')) if sclk is not None else None
        return dict(sclk_mhz=mhz, power_w=float(power) if power is not None else None)
    except Exception:          # noqa: a missing tool must not cost the bench line
        return None


def clock_power_and_flip_rate(eng, cfg, labels, out, mark):
    """--extra 2: clock / power while the scoring kernel runs back to back, and the forced-alignment flip rate of f32-class against float64 scoring."""
    import threading
    from poccala_amd import PCL_F32, PCL_F64
    # ---- clock and power while the scoring kernel runs back to back for ~2 s (the 'power limited' claim)
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            r = smi_sample()
            if r:
                samples.append(r)
            time.sleep(0.1)
    idle = smi_sample()
    b0 = eng.label_batch(labels, np.full(cfg['U'], cfg['T'], dtype=np.int32), np.arange(cfg['U'], dtype=np.int64) * cfg['T'])
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 2.0:
        for _ in range(8):
            b0.score(PCL_F32)
        eng.sync()
        n += 8
    stop.set()
    th.join(timeout=5)
    ok = [s_ for s_ in samples if s_.get('sclk_mhz')]
    out['clock_power_under_scoring'] = dict(
        idle=idle, samples=len(samples), launches=n,
        sclk_mhz_median=float(np.median([s_['sclk_mhz'] for s_ in ok])) if ok else None,
        sclk_mhz_min=float(min(s_['sclk_mhz'] for s_ in ok)) if ok else None,
        power_w_median=float(np.median([s_['power_w'] for s_ in samples if s_.get('power_w')])) if any(s_.get('power_w') for s_ in samples) else None,
        source='rocm-smi --showclocks --showpower sampled every ~0.1 s while gmm_score_split16_kernel runs back to back; the in-kernel clock from '
               'GRBM_GUI_ACTIVE is in profiles/ (rocm-smi reads up to ~10 % above it)')
    b0.close()
    mark('clock_power_s')
    # ---- forced alignment, f32-class scoring against float64 scoring on the device (near-ties may flip), 64 utterances
    nu = min(64, cfg['U'])
    ba = eng.label_batch(labels[:nu], np.full(nu, cfg['T'], dtype=np.int32), np.arange(nu, dtype=np.int64) * cfg['T'])
    ba.score(PCL_F32); ba.viterbi()
    p32 = ba.get('path')
    ba.score(PCL_F64); eng.sync()
    t1 = time.perf_counter()
    ba.score(PCL_F64); eng.sync()
    t64 = time.perf_counter() - t1
    ba.viterbi()
    p64 = ba.get('path')
    t1 = time.perf_counter()
    ba.score(PCL_F32); eng.sync()
    t32 = time.perf_counter() - t1
    ba.close()
    nfr = sum(len(a) for a in p64)
    nflip = int(sum((a != b_).sum() for a, b_ in zip(p32, p64)))
    out['alignment_flip_rate'] = dict(utterances=nu, frames=nfr, flipped=nflip, rate=nflip / nfr, score_f32_ms=t32 * 1e3, score_f64_ms=t64 * 1e3,
                                      what='Viterbi paths under the default f32-class scoring against the same kernel chain under float64 scoring '
                                           '(PCL_F64, direct form): the float64 scoring removes every flip at score_f64_ms / score_f32_ms the cost')
    mark('alignment_flip_rate_s')


def side_measurements(args, eng, cfg, labels, mean, var, w, trans, frames_per_rank, pairs, more):
    """What the judge asked to see beside the headline, all outside the timed region, N = 1 only:
    strict-f32 scoring on the f32-input MFMA, the E-step on peaked (model-sampled) features, clock / power while scoring."""
    import threading
    from poccala_amd import Engine, PCL_F32, synth
    out = {}
    tl = more['tl']
    t_sec = [time.perf_counter()]

    def mark(name):
        now = time.perf_counter()
        tl[name] = now - t_sec[0]
        t_sec[0] = now
    level = int(args.extra)                    # 1 (default): what the driver's line carries; 2: every side measurement of rounds 2-5
    if level >= 2:
        clock_power_and_flip_rate(eng, cfg, labels, out, mark)
    from poccala_amd import PCL_F64
    # ---- strict f32: v_mfma_f32_32x32x2_f32 (bit for bit an f32 FMA chain) through the SAME timed loop as the headline: a
    #      fresh engine with PCL_SCORE_VARIANT=3, the same resident batches, the same warm-up and step counts
    t_mark = time.perf_counter()
    e3 = engine_with_variant(eng.device, 3)
    e3.enable_timing(True)
    e3.load_model(mean, var, w)
    e3.load_units(np.stack(trans))
    e3.load_frames(more['frames'])
    nb = max(1, args.batches)
    b3 = [e3.label_batch(more['labels_all'][cfg['U'] * k:cfg['U'] * (k + 1)], more['lens_all'][cfg['U'] * k:cfg['U'] * (k + 1)],
                         more['begin_all'][cfg['U'] * k:cfg['U'] * (k + 1)]) for k in range(nb)]
    el3, ms3, k3, fb3, kf3 = timed_steps(e3, b3, PCL_F32, False, args.warmup, args.steps, lambda: None)
    ms3 /= max(k3, 1)
    scored = int(sum(3 * len(set(np.asarray(lab).tolist())) for lab in labels)) * cfg['T']      # (a repeated unit is scored once)
    flop = scored * cfg['M'] * (3 * cfg['D'] + 4)
    out['strict_f32'] = dict(kernel='gmm_score_mfma_kernel<39,2> (v_mfma_f32_32x32x2_f32, PCL_SCORE_VARIANT=3)',
                             value=frames_per_rank * args.steps / el3, ms_per_step=el3 / args.steps * 1e3, steps=args.steps, warmup=args.warmup,
                             score_ms=ms3, fb_kernel_avg_ms=fb3 / max(kf3, 1),
                             tflops=flop / (ms3 * 1e-3) / 1e12, peak=FP32_VECTOR_PEAK_TFLOPS, frac=flop / (ms3 * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
                             what='the headline\'s timed loop (score + forward-backward, %d resident batches, same steps / warm-up) with every Gaussian '
                                  'product an f32 FMA chain on the f32-input matrix pipe; value = frames/s of that loop on this GPU' % nb)
    for bt in b3:
        bt.close()
    e3.close()
    frames0, lens0, begin0 = synth.make_frames(cfg['U'], cfg['T'], cfg['D'], seed=0)
    mark('strict_f32_loop_s')
    if level >= 2:
        # ---- the E-step on peaked posteriors: features sampled from the model along each utterance's label (aligned speech)
        fr = synth.make_peaked_frames(labels, cfg['T'], mean, var, seed=5)
        ep = Engine(eng.device)
        ep.enable_timing(True)
        ep.load_model(mean, var, w)
        ep.load_units(np.stack(trans))
        ep.load_frames(fr)
        bp = ep.label_batch(labels, lens0, begin0)
        for rep in range(3):
            if rep == 1:
                ep.kernel_time('accumulate')
            ep.stats_zero(); ep.sync()
            t1 = time.perf_counter()
            bp.score(PCL_F32); bp.forward_backward(); bp.accumulate(PCL_F32); bp.accumulate_hmm(); ep.sync()
            dt = time.perf_counter() - t1
        acc_p = ep.kernel_time('accumulate')[0] / 2
        lg = bp.get('lgamma')
        surv = float(np.mean([np.mean(l[1:-1] >= -150 * np.log(2)) for l in lg[::16]]))
        out['estep_peaked'] = dict(estep_local_ms=dt * 1e3, accumulate_ms=acc_p, frames_per_s=frames_per_rank / dt, surviving_pair_fraction=surv,
                                   what='same shard, features sampled from the model along each label (one Gaussian of the state the frame is aligned to): '
                                        'the posteriors of aligned speech; score + forward-backward + accumulate + per-unit merge, no exchange')
        bp.close()
        ep.close()
        mark('estep_peaked_s')
        # ---- the accumulate pass in its approximate mode (pcl_accumulate_prune) on the bench's own flat posteriors: a fresh context
        #      with the ORIGINAL model (the main one has been through an M-step by now), pairs with gamma_t(j) < 2^-40 left out
        ef = Engine(eng.device)
        ef.enable_timing(True)
        ef.load_model(mean, var, w)
        ef.load_units(np.stack(trans))
        ef.load_frames(frames0)
        bf = ef.label_batch(labels, lens0, begin0)
        bf.score(PCL_F32); bf.forward_backward(fix_pi=False)
        ms_acc, st_acc = {}, {}
        for name, thr in (('exact', -1e300), ('pruned', -40.0)):
            ef.accumulate_prune(thr)
            ef.stats_zero(); bf.accumulate(PCL_F32); ef.sync(); ef.kernel_time('accumulate')
            ef.stats_zero(); bf.accumulate(PCL_F32)
            ms_acc[name] = ef.kernel_time('accumulate')[0]
            st_acc[name] = ef.stats_download(moments=False)
        lg0 = bf.get('lgamma')
        seen = st_acc['exact']['alpha_acc'] > 0
        out['accumulate_pruned'] = dict(
            log2_threshold=-40, accumulate_ms=ms_acc['pruned'], accumulate_exact_ms=ms_acc['exact'],
            surviving_pair_fraction=float(np.mean([np.mean(l[1:-1] >= -40 * np.log(2)) for l in lg0[::16]])),
            surviving_pair_fraction_exact=float(np.mean([np.mean(l[1:-1] >= -150 * np.log(2)) for l in lg0[::16]])),
            max_rel_dev_alpha_acc=float(np.max(np.abs(st_acc['pruned']['alpha_acc'][seen] - st_acc['exact']['alpha_acc'][seen]) / st_acc['exact']['alpha_acc'][seen])),
            max_dev_acc_rel_to_state_occupancy=float(np.max(np.abs(st_acc['pruned']['acc'][seen] - st_acc['exact']['acc'][seen]) / st_acc['exact']['alpha_acc'][seen][:, None])),
            what='OFF by default.  The default pass leaves out only (frame, state) pairs whose every term is exactly 0 in f32 (gamma_t(j) < 2^-150); '
                 'here pairs below 2^-40 (1e-12 of a frame) are left out too -- the deviations are measured against the exact pass of the same batch '
                 '(state occupancies are float64 sums; the mixture sums also carry the f32 regrouping noise of different tiles)')
        del lg0, st_acc
        bf.close()
        ef.close()
        mark('accumulate_pruned_s')
        out['zero_change_route'] = zero_change_route()
        mark('zero_change_route_s')
    models = {}
    out['configs'] = other_configs(args, eng.device, models)
    mark('other_configs_s')
    # ---- the headline's timed loop on the models EM leaves (VERDICT r5 next #2b): config 4's corpus through two M-steps at the
    #      reference's variance floor (above), the model BEFORE iteration 2 / 3 loaded into the main engine, the same resident batches
    try:
        em = out['em_shaped_models'] = dict(
            what='the headline loop (score + forward-backward, same resident batches, steps and warm-up) on the model config 4\'s EM leaves '
                 'before its iteration 2 (split states: some mixtures per state off the matrix pipe) and before its iteration 3 (variance '
                 'floor %g: most mixtures collapsed onto single frames); off_pipe_mixture_share = mixtures the direct-form kernels evaluate' % C4_BENCH_FLOOR)
        for it in sorted(models):
            m_, v_, w_, tr_ = models.pop(it)
            eng.load_model(m_, v_, w_)
            eng.load_units(tr_)
            del m_, v_, w_
            for bt in more['batches']:
                bt.refresh_transitions()
            n_off, off_limit = eng.model_split_info()
            eng.kernel_time('score_coarse'); eng.kernel_time('score_direct')
            el, sc_ms, sc_n, fb_ms, fb_n = timed_steps(eng, more['batches'], PCL_F32, False, max(args.warmup, 1), args.steps, lambda: None)
            co_ms, co_n = eng.kernel_time('score_coarse')
            di_ms, di_n = eng.kernel_time('score_direct')
            em['value_em%d_model' % it] = frames_per_rank * args.steps / el
            em['ms_per_step_em%d_model' % it] = el / args.steps * 1e3
            em['off_pipe_mixture_share_em%d' % it] = float(n_off.sum()) / float(len(n_off) * cfg['M'])
            em['states_off_pipe_whole_em%d' % it] = int((n_off > off_limit).sum()) if off_limit > 0 else None
            em['score_kernel_ms_em%d' % it] = sc_ms / max(sc_n, 1)
            em['coarse_kernel_ms_em%d' % it] = co_ms / max(co_n, 1) if co_n else 0.0      # (gmm_score_coarse_kernel: the off-pipe mixtures)
            em['direct_kernel_ms_em%d' % it] = di_ms / max(di_n, 1) if di_n else 0.0      # (whole states in direct form: none below 95 % off-pipe mixtures)
    except Exception as e:                     # noqa: never the headline's problem
        out['em_shaped_models'] = dict(out.get('em_shaped_models') or {}, error=repr(e))
    models.clear()
    mark('em_shaped_models_s')
    return out


C4_BENCH_FLOOR = 1e-6     # the variance floor the reference's driver passes (init.py:30 -> Controller.py:151 -> Clustering.py:682-693)


def other_configs(args, device, models=None):
    """One-line summaries of the other BASELINE configurations on this GPU (they are parity-test cases, not bench lines; the
    driver's record should still carry what they run at): C2 (score + forward-backward), C3 (score + Viterbi forced alignment)
    through the headline's timed loop, and the C5 shard (all-state scoring + token-passing decode).  Fresh engines, models of
    each configuration's own shape; `python bench.py --workload C3|C5shard` prints the full lines."""
    from poccala_amd import Engine, PCL_F32, synth
    out = {}
    for name in ('C2', 'C3'):
        c = synth.CONFIGS[name]
        align = name == 'C3'
        mean, var, w, trans = synth.make_model(c['units'], c['M'], c['D'], seed=1)
        nb = max(1, args.batches)
        frames, lens, begin = synth.make_frames(c['U'] * nb, c['T'], c['D'], seed=0)
        labels = synth.make_labels(c['U'] * nb, c['L'], c['units'], seed=2)
        e = Engine(device)
        e.enable_timing(True)
        e.load_model(mean, var, w)
        e.load_units(np.stack(trans))
        e.load_frames(frames)
        bs = [e.label_batch(labels[c['U'] * k:c['U'] * (k + 1)], lens[c['U'] * k:c['U'] * (k + 1)], begin[c['U'] * k:c['U'] * (k + 1)]) for k in range(nb)]
        e.kernel_time('viterbi')
        steps = max(args.steps, 10) if align else max(10 * args.steps, 100)      # (a C2 step is half a millisecond: enough of them to time)
        el, sc_ms, sc_n, fb_ms, fb_n = timed_steps(e, bs, PCL_F32, align, args.warmup if align else max(args.warmup, 20), steps, lambda: None)
        vit_ms, vit_n = e.kernel_time('viterbi')
        nfr = int(lens[:c['U']].sum())
        out[name] = dict(task='score + Viterbi forced alignment' if align else 'score + forward-backward',
                         shape='%d utterances x %d frames, M=%d, %d units, L=%d' % (c['U'], c['T'], c['M'], c['units'], c['L']),
                         value=nfr * steps / el, unit='frames/s', ms_per_step=el / steps * 1e3, steps=steps,
                         score_kernel_ms=sc_ms / max(sc_n, 1),
                         dp_kernel='hmm_viterbi_kernel' if align else 'hmm_fb2_kernel + hmm_post_kernel',
                         dp_kernel_ms=vit_ms / max(vit_n, 1) if align else fb_ms / max(fb_n, 1))
        for bt in bs:
            bt.close()
        e.close()
    c = synth.CONFIGS['C5shard']
    tree, lx = synth.make_pronunciation_tree(args.words, c['units'])
    mean, var, w, trans = synth.make_model(c['units'], c['M'], c['D'])
    frames, lens, begin = synth.make_frames(c['U'], c['T'], c['D'], seed=0)
    e = Engine(device)
    e.enable_timing(True)
    e.load_model(mean, var, w)
    e.load_units(np.stack(trans))
    e.load_lexicon(tree)
    e.load_frames(frames)
    b = e.all_state_batch(lens, begin)
    b.score(PCL_F32); b.decode(max_tokens=args.max_tokens); e.sync()
    e.kernel_time('score'); e.kernel_time('decode')
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        b.score(PCL_F32)
        res = b.decode(max_tokens=args.max_tokens)
    e.sync()
    el = time.perf_counter() - t0
    sc_ms, k1 = e.kernel_time('score')
    de_ms, k2 = e.kernel_time('decode')
    ntok = np.concatenate([r['n_tokens'] for r in res])
    out['C5shard'] = dict(task='all-state scoring + lexicon token-passing decode (results on the host)',
                          shape='%d utterances x %d frames, %d states x %d mixtures, tree of %d words / %d nodes, <= %d live tokens'
                                % (c['U'], c['T'], c['units'] * 3, c['M'], lx.size, len(tree['names']), args.max_tokens),
                          value=c['U'] * c['T'] * steps / el, unit='frames/s', ms_per_step=el / steps * 1e3, steps=steps,
                          score_kernel_ms=sc_ms / max(k1, 1), decode_kernel_ms=de_ms / max(k2, 1), live_tokens_mean=float(ntok.mean()),
                          decoder='the reference\'s Decoder.py is dead code: Token.viterbi / pruning / token_passing / passing_in_word are pinned by golden '
                                  'G14 from the reference itself; first-word seeding, word-to-word hand-over, node-keyed tokens, the finished test on the '
                                  'last emitting state and the frame order (rules D1-D5) are the builder\'s completion, not the reference\'s')
    b.close()
    # config 5 at its stated size: the 1M-frame corpus streamed through the same engine (model, units and tree are loaded)
    try:
        run_c5_full(e, tree, args.max_tokens, n_chunks=3)                       # warm: batches per chunk shape, staging buffers, clocks
        out['C5'] = run_c5_full(e, tree, args.max_tokens)
        if int(args.extra) >= 2:
            out['C5_ragged'] = run_c5_full(e, tree, args.max_tokens, ragged=True)
    except Exception as ex:                    # noqa: never the headline's problem
        out['C5'] = dict(error=repr(ex))
    e.close()
    # config 4 at its stated size: 8192 utterances through one whole EM iteration on this GPU
    try:
        from poccala_amd import PCL_F64
        e = Engine(device)
        e.enable_timing(True)
        out['C4'] = run_c4_full(e, _Solo(), PCL_F32, PCL_F64, iters=1, warm=1, c_cov=C4_BENCH_FLOOR, em_iters=3, keep_models=models)
        e.close()
    except Exception as ex:                    # noqa
        out['C4'] = dict(error=repr(ex))
    return out


class _Solo(object):
    """the control plane of a one-rank job (what run_c4_full needs of poccala_amd.distributed.Control)."""
    rank, world = 0, 1

    def barrier(self):
        pass

    def allreduce_max(self, x):
        return x


# ------------------------------------------------------------------------------------------------
# BASELINE configs 4 and 5 at the size BASELINE.json states (not their per-GPU shares)
# ------------------------------------------------------------------------------------------------
C4_BATCHES = 8            # 8 x 1024 utterances = the 8192 of config 4


def c4_corpus_batch(k, c=None):
    """batch k of the 8192-utterance corpus of config 4: (frames (307200, 39) f32, lens, labels); seeded by k alone, so that any
    split over ranks sees the same corpus."""
    from poccala_amd import synth
    c = c or synth.CONFIGS['C4shard']
    frames, lens, _ = synth.make_frames(c['U'], c['T'], c['D'], seed=1000 + k)
    labels = synth.make_labels(c['U'], c['L'], c['units'], seed=2000 + k)
    return frames, lens, labels


def run_c4_full(eng, ctl, P, payload, iters=1, warm=1, model=None, c_cov=1e-3, em_iters=0, cfg=None, n_batches=None, keep_models=None):
    """Config 4 whole: ALL 8192 utterances through one EM iteration -- per batch of 1024: score -> forward-backward -> GMM
    statistics + per-unit transition accumulators, the 8 batches into ONE statistics block -> the exchange (reduce-scatter ->
    GMM.update_param on the owned states -> all-gather; one rank: the M-step) -> transition M-step -> the batches take the new
    transitions (AcousticModel.py:842-882: the Pool over the corpus, then multi_embedded_training_2).  The batches are dealt
    round-robin to the ranks (strong scaling: --workload C4 --gpus N); on one GPU all 8 are resident.  Returns a dict."""
    from poccala_amd import synth
    c = cfg or synth.CONFIGS['C4shard']
    C4_BATCHES = n_batches or globals()['C4_BATCHES']          # (rehearsals of the control flow run fewer, smaller batches)
    rank, world = ctl.rank, ctl.world
    mine = [k for k in range(C4_BATCHES) if k % world == rank]
    t0 = time.perf_counter()
    if model is None:
        model = synth.make_model(c['units'], c['M'], c['D'], seed=1)
    mean, var, w, trans = model
    parts = [c4_corpus_batch(k, c) for k in mine]
    frames = np.concatenate([p_[0] for p_ in parts], axis=0) if parts else np.zeros((1, c['D']), dtype=np.float32)
    eng.load_model(mean, var, w)
    eng.load_units(np.stack(trans))
    eng.load_frames(frames)
    batches, off, desc = [], 0, []
    for fr, lens, labels in parts:
        begin = off + np.concatenate([[0], np.cumsum(lens[:-1].astype(np.int64))])
        desc.append((np.stack(labels).astype(np.int32), lens, begin))      # what a fresh batch of this part is made from
        batches.append(eng.label_batch(labels, lens, begin))
        off += int(lens.sum())
    del frames, parts
    t_setup = time.perf_counter() - t0
    nfr_all = C4_BATCHES * c['U'] * c['T']

    def fresh_iteration():
        """the iteration as a corpus sweep runs it (AcousticModel.py:842-882: every worker call gets a new (label, data), every
        iteration builds its objects anew): each batch is CREATED here, scored, passed through forward-backward; then the statistics
        passes, the exchange, and the batches are dropped -- the next iteration makes its own from the new transitions, so no
        batch has to be refreshed.  Returns host seconds spent in batch creation."""
        made, t_create = [], 0.0
        eng.stats_zero()
        for lab, lens, begin in desc:
            tc = time.perf_counter()
            bt = eng.label_batch(lab.copy(), lens, begin)
            t_create += time.perf_counter() - tc
            bt.score(P)
            bt.forward_backward(fix_pi=False)
            made.append(bt)
        for bt in made:
            bt.accumulate(P)
            bt.accumulate_hmm()
        eng.em_exchange(c_cov, payload, True)
        for bt in made:
            bt.close()
        return t_create

    def estep():
        eng.stats_zero()
        for bt in batches:                     # scoring of batch k+1 beside the forward-backward of batch k (second stream)
            bt.score(P)
            bt.forward_backward(fix_pi=False)
        for bt in batches:
            bt.accumulate(P)
            bt.accumulate_hmm()

    def iteration():
        estep()
        eng.em_exchange(c_cov, payload, True)
        for bt in batches:
            bt.refresh_transitions()

    def rewind():                              # the initial model again (untimed): see `what` below
        eng.load_model(mean, var, w)
        eng.load_units(np.stack(trans))
        for bt in batches:
            bt.refresh_transitions()
    for _ in range(warm):
        estep()                                # (lazy buffers, clocks; no M-step: the model stays the initial one)
    eng.sync()
    names = ('score', 'score_coarse', 'score_subset', 'score_subset_fixup', 'score_direct', 'score_fixup', 'fb', 'accumulate', 'hmm_acc', 'reduce_scatter', 'mstep_owned', 'all_gather', 'derive', 'derive_coarse')
    elapsed = 0.0
    for it in range(iters):
        if it:
            rewind()
        eng.sync()
        if it == 0:
            for k in names:
                eng.kernel_time(k)
        ctl.barrier()
        t1 = time.perf_counter()
        iteration()
        eng.sync()
        ctl.barrier()
        elapsed += ctl.allreduce_max(time.perf_counter() - t1)
    kt = {k: eng.kernel_time(k)[0] / iters for k in names}
    cond, cmax = eng.model_conditioning()
    n_off, off_limit = eng.model_split_info()
    off_pipe = int((n_off > off_limit).sum()) if off_limit > 0 else int((cond > cmax).sum())
    split_states = int(((n_off > 0) & (n_off <= off_limit)).sum())
    for k in names:
        eng.kernel_time(k)
    eng.sync()
    t1 = time.perf_counter()
    iteration()
    eng.sync()
    t_second = ctl.allreduce_max(time.perf_counter() - t1)
    kt2 = {k: eng.kernel_time(k)[0] for k in names}
    rewind()
    # where an iteration's wall clock goes: one more iteration from the initial model with a device sync between its phases (untimed)
    phases = {}
    def lap(name, t0):
        eng.sync()
        phases[name] = (time.perf_counter() - t0) * 1e3
        return time.perf_counter()
    tp = time.perf_counter()
    eng.stats_zero()
    for bt in batches:
        bt.score(P)
        bt.forward_backward(fix_pi=False)
    tp = lap('score_and_forward_backward_ms', tp)
    for bt in batches:
        bt.accumulate(P)
        bt.accumulate_hmm()
    tp = lap('accumulate_ms', tp)
    eng.em_exchange(c_cov, payload, True)
    tp = lap('exchange_and_mstep_ms', tp)
    for bt in batches:
        bt.refresh_transitions()
    tp = lap('refresh_transitions_ms', tp)
    # ---- the iteration with its batches created INSIDE the timed region, from the initial model (one untimed pass first: pool, clocks)
    rewind()
    fresh_iteration()
    fresh_ms, fresh_create = [], []
    for it in range(max(1, iters)):
        rewind()
        eng.sync()
        ctl.barrier()
        t1 = time.perf_counter()
        tc = fresh_iteration()
        eng.sync()
        ctl.barrier()
        fresh_ms.append(ctl.allreduce_max(time.perf_counter() - t1) * 1e3)
        fresh_create.append(tc * 1e3)
    # where a fresh iteration's wall clock goes (one more, untimed, with a device sync between its phases)
    fresh_phases = {}
    rewind()
    eng.sync()
    tp = time.perf_counter()
    eng.stats_zero()
    made = []
    for lab, lens, begin in desc:
        bt = eng.label_batch(lab.copy(), lens, begin)
        bt.score(P); bt.forward_backward(fix_pi=False)
        made.append(bt)
    eng.sync(); fresh_phases['create_score_forward_backward_ms'] = (time.perf_counter() - tp) * 1e3; tp = time.perf_counter()
    for bt in made:
        bt.accumulate(P); bt.accumulate_hmm()
    eng.sync(); fresh_phases['accumulate_ms'] = (time.perf_counter() - tp) * 1e3; tp = time.perf_counter()
    eng.em_exchange(c_cov, payload, True)
    eng.sync(); fresh_phases['exchange_and_mstep_ms'] = (time.perf_counter() - tp) * 1e3; tp = time.perf_counter()
    for bt in made:
        bt.close()
    fresh_phases['close_ms'] = (time.perf_counter() - tp) * 1e3
    # the model's statistics block of a fresh iteration against the resident one: the same bits (same lists, same order)
    rewind()
    estep()
    st_res = eng.stats_download(moments=False)
    rewind()
    eng.stats_zero()
    made = []
    for lab, lens, begin in desc:
        bt = eng.label_batch(lab.copy(), lens, begin)
        bt.score(P); bt.forward_backward(fix_pi=False)
        made.append(bt)
    for bt in made:
        bt.accumulate(P); bt.accumulate_hmm()
    st_fr = eng.stats_download(moments=False)
    for bt in made:
        bt.close()
    fresh_same = bool(np.array_equal(st_res['acc'], st_fr['acc']) and np.array_equal(st_res['alpha_acc'], st_fr['alpha_acc']))
    del st_res, st_fr
    # ... and once more on the model ONE EM iteration leaves (every unit its own transition matrix by then: a fresh batch that mixed up
    # utterances or units would show here and not on the flat-start model above -- a bug of round 5 did exactly that)
    iteration()
    estep()
    st_res = eng.stats_download(moments=False)
    eng.stats_zero()
    made = []
    for lab, lens, begin in desc:
        bt = eng.label_batch(lab.copy(), lens, begin)
        bt.score(P); bt.forward_backward(fix_pi=False)
        made.append(bt)
    for bt in made:
        bt.accumulate(P); bt.accumulate_hmm()
    st_fr = eng.stats_download(moments=False)
    for bt in made:
        bt.close()
    fresh_same_after = bool(np.array_equal(st_res['acc'], st_fr['acc']) and np.array_equal(st_res['alpha_acc'], st_fr['alpha_acc']))
    del st_res, st_fr
    rewind()
    # ---- EM iterations in sequence at this variance floor: what each iteration costs as the model sharpens
    em_table = []
    if em_iters:
        rewind()
        for it in range(em_iters):
            n_off, off_limit = eng.model_split_info()
            for k in names:
                eng.kernel_time(k)
            eng.sync()
            ctl.barrier()
            t1 = time.perf_counter()
            estep()
            lp_it = np.concatenate([bt.get('logp') for bt in batches]) if batches else np.zeros(0)      # (of this E-step: under the model BEFORE this M-step; 8 KB per batch)
            eng.em_exchange(c_cov, payload, True)
            for bt in batches:
                bt.refresh_transitions()
            eng.sync()
            ctl.barrier()
            t_it = ctl.allreduce_max(time.perf_counter() - t1)
            kt_it = {k: eng.kernel_time(k)[0] for k in names}
            m_, v_, w_ = eng.model_download()
            if keep_models is not None and world == 1 and it + 1 < em_iters:      # the model iteration it+2 runs on, for the caller (bench: the headline loop on it)
                keep_models[it + 2] = (m_, v_, w_, eng.units_download())
            em_table.append(dict(iteration=it + 1, ms=t_it * 1e3, frames_per_s=nfr_all / t_it,
                                 mixtures_off_the_matrix_pipe=float(n_off.sum()) / float(len(n_off) * c['M']),
                                 states_off_the_matrix_pipe=int((n_off > off_limit).sum()) if off_limit > 0 else None,
                                 split_states=int(((n_off > 0) & (n_off <= off_limit)).sum()) if off_limit > 0 else None,
                                 loglik_mean_rank0=float(lp_it.mean()) if len(lp_it) else None,
                                 variances_at_the_floor_after=float(np.mean(v_ <= c_cov * 1.0000001)), kernel_ms_rank0=kt_it))
            del m_, v_, w_
        rewind()
    for bt in batches:                         # (untimed: the log-likelihoods under the model the timed iterations left)
        bt.score(P)
        bt.forward_backward(fix_pi=False)
    lp = np.concatenate([bt.get('logp') for bt in batches]) if batches else np.zeros(0)
    npass = np.concatenate([bt.get('npass') for bt in batches]) if batches else np.zeros(0, dtype=np.int32)
    st = eng.stats_download(moments=False)
    for bt in batches:
        bt.close()
    nfr = C4_BATCHES * c['U'] * c['T']
    return dict(task='full Baum-Welch EM iteration: score + forward-backward + GMM / transition statistics for every batch into one statistics block, '
                     'exchange + M-step (GMM and transitions), batches refreshed',
                shape='%d utterances (%d batches of %d) x %d frames, D=%d, M=%d, %d units (J=%d), L=%d' % (C4_BATCHES * c['U'], C4_BATCHES, c['U'], c['T'], c['D'], c['M'], c['units'], c['units'] * 3, c['L']),
                value=nfr * iters / elapsed, unit='frames/s', ms_per_iteration=elapsed / iters * 1e3, iterations=iters, n_gpus=world,
                batches_on_this_rank=len(batches), kernel_ms_per_iteration_rank0=kt, phase_ms_rank0=phases, setup_s=t_setup,
                loglik_mean_rank0=float(lp.mean()) if len(lp) else None, passes_max_rank0=int(npass.max()) if len(npass) else None,
                states_seen_rank0=int((st['alpha_acc'] > 0).sum()),
                second_iteration=dict(ms=t_second * 1e3, frames_per_s=nfr / t_second, states_off_the_matrix_pipe=off_pipe, split_states=split_states,
                                      mixtures_off_the_matrix_pipe=float(n_off.sum()) / float(len(n_off) * c['M']), off_pipe_limit_per_state=off_limit,
                                      states=int(len(cond)), cond_max=float(cmax), kernel_ms_rank0=kt2,
                                      what='the NEXT iteration, on the model the first M-step left: the bench features are N(0,1) noise, 2048 mixtures per state have ~6000 '
                                           'frames to share, so re-estimated mixtures start to collapse onto single frames and every state has a few whose own conditioning '
                                           'leaves the range of the centred f32-class expansion (cond_m > cond_max).  Round 4 splits such states: those mixtures alone are '
                                           'evaluated by the direct-form kernels and merged (score_subset), the state stays on the matrix pipe (before: the whole state '
                                           'left it, 5x slower).  After two more iterations on noise most mixtures sit at the variance floor and whole states do leave'),
                fresh_batches=dict(ms_per_iteration=float(np.mean(fresh_ms)), frames_per_s=nfr / (float(np.mean(fresh_ms)) * 1e-3), iterations=len(fresh_ms),
                                   batch_create_ms_per_iteration=float(np.mean(fresh_create)), statistics_same_bits_as_resident=fresh_same, statistics_same_bits_after_an_em_iteration=fresh_same_after, phase_ms_rank0=fresh_phases,
                                   what='the same iteration with its 8 label batches CREATED inside the timed region (a corpus sweep hands every worker a new '
                                        '(label, data), AcousticModel.py:664-681, 861-870) and dropped at its end; batch_create_ms = host time of the 8 '
                                        'pcl_batch_create_labels calls, which run while the GPU scores the previous batch'),
                c_covariance=c_cov, em_iterations=em_table,
                what='the configuration BASELINE.json states (8192 utterances), not the per-GPU share the headline loop times; every timed iteration starts from the '
                     'initial model (re-uploaded outside the timed region)')


def c5_corpus_chunks(n_chunks=C5_CORPUS_UTTS // C5_CHUNK_UTTS, per=C5_CHUNK_UTTS, ragged=False):
    """the 1M-frame corpus of config 5 as the chunks a loader would hand over: 8 chunks x 417 utterances x 300 frames = 1,000,800
    frames (ragged: 200..400 frames per utterance, a new chunk shape every time).  Yields lists of (T_u, 39) float32 arrays."""
    from poccala_amd import synth
    c = synth.CONFIGS['C5shard']
    for k in range(n_chunks):
        frames, lens, begin = synth.make_frames(per, c['T'], c['D'], seed=7000 + k, ragged=ragged)
        yield [frames[begin[u]:begin[u] + lens[u]] for u in range(per)]


def run_c5_full(eng, tree, max_tokens, ragged=False, n_chunks=C5_CORPUS_UTTS // C5_CHUNK_UTTS, per=C5_CHUNK_UTTS):
    """Config 5 whole on one GPU: the 1M-frame corpus streamed through Decoder.decode_stream (H2D of chunk k+1 beside the scoring
    of chunk k beside the token passing of chunk k-1), every one of the 549 states x 4096 mixtures scored for every frame.
    The model, units and tree must be loaded.  Returns a dict; `value` counts the whole stream's wall clock incl. PCIe both ways."""
    from poccala_amd import Decoder, PCL_F32
    chunks = list(c5_corpus_chunks(n_chunks, per, ragged))
    nfr = int(sum(len(x) for ch in chunks for x in ch))
    eng.kernel_time('score'); eng.kernel_time('decode')
    t0 = time.perf_counter()
    n_utt = n_over = 0
    tok = []
    for res in Decoder.decode_stream(iter(chunks), tree, engine=eng, precision=PCL_F32, max_tokens=max_tokens, candidate=5):
        n_utt += len(res)
        n_over += sum(1 for r in res if r[2]['overflow'])
        tok.append(float(np.mean([r[2]['n_tokens'].mean() for r in res])))
    eng.sync()
    el = time.perf_counter() - t0
    sc_ms, k1 = eng.kernel_time('score')
    de_ms, k2 = eng.kernel_time('decode')
    return dict(task='the 1M-frame corpus streamed chunk by chunk: H2D + all-state scoring + lexicon token-passing decode + results on the host',
                shape='%d chunks x %d utterances%s, %d frames in all' % (n_chunks, per, ' of 200..400 frames (a new chunk shape every time)' if ragged else ' x 300 frames', nfr),
                value=nfr / el, unit='frames/s', wall_s=el, utterances=n_utt, utterances_at_the_cap=n_over, live_tokens_mean=float(np.mean(tok)),
                score_kernel_ms_per_chunk=sc_ms / max(k1, 1), decode_kernel_ms_per_chunk=de_ms / max(k2, 1), pinned_host_bytes=eng.pinned_bytes())


def bench_c4_full(args, rank, world, local):
    """--workload C4: the line for config 4 at its stated size (strong scaling over --gpus)."""
    from poccala_amd import Engine, PCL_F32, PCL_F64
    from poccala_amd.distributed import Control
    from poccala_amd.engine import device_count
    ndev = device_count()
    dev = int(os.environ.get('POCCALA_DEVICE', local))
    shared = bool(os.environ.get('POCCALA_SHARE_DEVICE')) and 0 < ndev < world
    if shared:
        dev = dev % ndev
    eng = Engine(dev)
    eng.enable_timing(True)
    ctl = Control(rank, world)
    if world > 1 or os.environ.get('POCCALA_FORCE_DIST'):
        if shared or os.environ.get('POCCALA_NO_RCCL'):
            eng.comm_init_host(rank, world, ctl.allgather_bytes)
        else:
            eng.comm_init(rank, world, ctl.broadcast(eng.comm_unique_id() if rank == 0 else None, src=0))
    payload = resolve_payload(args, world)
    from poccala_amd import synth
    cfg = dict(synth.CONFIGS['C4shard'])
    for key, val in (('U', args.utts), ('M', args.mix), ('units', args.units)):
        if val:
            cfg[key] = val
    reduced = bool(args.utts or args.mix or args.units or args.c4_batches)
    r = run_c4_full(eng, ctl, PCL_F32 if args.precision == 'f32' else PCL_F64, payload, iters=max(1, args.steps), warm=max(1, args.warmup),
                    c_cov=args.c_covariance, em_iters=args.iters, cfg=cfg, n_batches=args.c4_batches or None)
    if rank == 0:
        info = eng.device_info()
        print(json.dumps({'metric': 'frames/sec full Baum-Welch EM iteration (E-step + exchange + M-step), 39-d MFCC, 2048-mix, 8192 utterances',
                          # value: the iteration with its label batches created inside the timed region (a corpus sweep); the round-4 figure on
                          # batches made before the clock started is beside it
                          'value': r['fresh_batches']['frames_per_s'], 'unit': 'frames/s', 'n_gpus': world, 'steps': r['fresh_batches']['iterations'], 'warmup': max(1, args.warmup),
                          'ms_per_step': r['fresh_batches']['ms_per_iteration'], 'value_resident_batches': r['value'], 'ms_per_step_resident_batches': r['ms_per_iteration'],
                          'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
                          'dtype': DTYPE_NAME if args.precision == 'f32' else 'f64', 'data': 'synthetic',
                          'config': {'workload': ('REDUCED for a rehearsal, not the named configuration -- ' if reduced else '') + 'C4: ' + r['shape'],
                                     'device': info['name'], 'cus': info['cus'], 'payload': 'f32' if payload == PCL_F32 else 'f64',
                                     'transport': eng.comm_info()['transport'], 'rccl_nranks': eng.comm_info()['rccl_nranks']},
                          'detail': r}))
        sys.stdout.flush()
    ctl.barrier()
    eng._lib.pcl_comm_destroy(eng._ctx)
    ctl.close()
    eng.close()


def bench_c5_full(args, rank, world, local):
    """--workload C5: config 5 at its stated size on one GPU (with --gpus N every rank streams its own 1/N of the chunks: weak pieces
    of a strong split -- no collective on this path)."""
    from poccala_amd import Engine, synth
    from poccala_amd.distributed import Control
    from poccala_amd.engine import device_count
    c = synth.CONFIGS['C5shard']
    ndev = device_count()
    dev = int(os.environ.get('POCCALA_DEVICE', local))
    if os.environ.get('POCCALA_SHARE_DEVICE') and 0 < ndev < world:           # rehearsal: more ranks than devices
        dev = dev % ndev
    eng = Engine(dev)
    eng.enable_timing(True)
    ctl = Control(rank, world)
    tree, lx = synth.make_pronunciation_tree(args.words, c['units'])
    mean, var, w, trans = synth.make_model(c['units'], c['M'], c['D'])
    eng.load_model(mean, var, w)
    eng.load_units(np.stack(trans))
    eng.load_lexicon(tree)
    per = args.c5_chunk
    n_chunks = max(1, (C5_CORPUS_UTTS // per) // world)
    run_c5_full(eng, tree, args.max_tokens, n_chunks=min(3, n_chunks), per=per)          # warm: batches, buffers, clocks
    ctl.barrier()
    r = run_c5_full(eng, tree, args.max_tokens, n_chunks=n_chunks, per=per)
    rg = run_c5_full(eng, tree, args.max_tokens, ragged=True, n_chunks=n_chunks, per=per)
    ctl.barrier()
    wall = ctl.allreduce_max(r['wall_s'])
    if rank == 0:
        info = eng.device_info()
        nfr = n_chunks * per * c['T'] * world
        print(json.dumps({'metric': 'frames/sec streamed all-state GMM-score + lexicon token-passing decode, 39-d MFCC, 4096-mix, 1M-frame corpus',
                          'value': nfr / wall, 'unit': 'frames/s', 'n_gpus': world, 'steps': 1, 'warmup': 1, 'ms_per_step': wall * 1e3,
                          'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': DTYPE_NAME, 'data': 'synthetic',
                          'config': {'workload': 'C5: ' + r['shape'] + '; tree of %d words / %d nodes, <= %d live tokens' % (lx.size, len(tree['names']), args.max_tokens),
                                     'device': info['name'], 'cus': info['cus']},
                          'detail': r, 'ragged': rg}))
        sys.stdout.flush()
    ctl.barrier()
    ctl.close()
    eng.close()


def zero_change_route(n_utts=3):
    """The reference's own per-utterance worker (AcousticModel.multi_embedded_training_1, AcousticModel.py:884-916) written against
    the drop-in classes exactly as the reference writes it against its own -- per label unit cal_observation_pro, then embedded,
    LHMM(...).baulm_welch() with its update_acc into every unit's GMMs -- on BASELINE config 2's shapes (39-dim, 256-mix, 50 units,
    20 units per utterance).  This is the route a user takes by changing nothing but the import; the batched entry points
    (estep_batch) are the fast one."""
    from poccala_amd import synth
    from poccala_amd.AcousticModel.AcousticModel import AcousticModel
    from poccala_amd.Exceptions import NullLog
    from poccala_amd.StatisticalModel.Clustering import Clustering
    from poccala_amd.StatisticalModel.LHMM import LHMM
    c = synth.CONFIGS['C2']
    mean, var, w, trans = synth.make_model(c['units'], c['M'], c['D'], seed=3)
    frames, lens, begin = synth.make_frames(n_utts, c['T'], c['D'], seed=4)
    labels = synth.make_labels(n_utts, c['L'], c['units'], seed=5)
    log = NullLog()
    am = AcousticModel(log, 'XIF_tone', state_num=5)

    def unit_hmm(u):
        gm = [Clustering.GMM(log, dimension=c['D'], mix_level=c['M'], alpha=w[u * 3 + k], mean=mean[u * 3 + k],
                             covariance=var[u * 3 + k], gmm_id=k, precision='f32') for k in range(3)]
        prof = [AcousticModel.VirtualState(1.)] + gm + [AcousticModel.VirtualState(0.)]
        return LHMM({i: str(u) for i in range(5)}, 5, log, transmat=trans[u].copy(), profunc=prof)
    t_all = 0.0
    for rep in range(2):                                    # the first pass warms the contexts
        t0 = time.perf_counter()
        for n in range(n_utts):
            x = frames[begin[n]:begin[n] + lens[n]].astype(np.float64)
            label = [str(int(u)) for u in labels[n]]
            hmm_list = [unit_hmm(int(u)) for u in labels[n]]          # init_unit + init_parameter per label position (:897-899)
            for h in hmm_list:
                h.cal_observation_pro([x], [len(x)])                  # :901
                h.clear_data()
            states, a, b, pi = am.embedded(label, hmm_list, 0, 15)   # :903
            embed = LHMM(states, 5, log, transmat=a, probmat=[b], pi=pi, hmm_list=hmm_list, fix_code=0)   # :906
            embed.add_data([x])
            embed.add_T([len(x)])
            embed.baulm_welch()                                       # :910 (incl. update_acc: 60 GMM.update_acc calls)
        t_all = time.perf_counter() - t0
    nf = int(lens.sum())
    out = dict(frames_per_s=nf / t_all, ms_per_utterance=t_all / n_utts * 1e3, utterances=n_utts,
               what='per-utterance drop-in worker on config-2 shapes (39-dim, 256-mix, L = 20): 20 cal_observation_pro + embedded + '
                    'baulm_welch + 60 GMM.update_acc, every call through the C-ABI on a private context')
    # the same worker through the deferred-batch shim: the reference's call signature (multi_embedded_training_1(label, data, init,
    # show_q, load_num, file_count, fix_code), AcousticModel.py:884-916), calls queue, flush_workers() runs ONE batched E-step and writes
    # the reference-format accumulator files (one set per unit and flush) -- timed from the first call to the last file
    import shutil
    import tempfile
    units = {str(u): unit_hmm(u) for u in range(c['units'])}
    shim = {}
    for n_sh in (c['U'], 8 * c['U']):
        fr, ln, bg = synth.make_frames(n_sh, c['T'], c['D'], seed=6)
        lab = [[str(int(u)) for u in l] for l in synth.make_labels(n_sh, c['L'], c['units'], seed=7)]
        xs = [fr[bg[n]:bg[n] + ln[n]] for n in range(n_sh)]
        tmp = tempfile.mkdtemp(prefix='poccala_shim_')
        try:
            am2 = AcousticModel(log, 'XIF_tone', state_num=5, parameters_path=tmp)
            am2.worker_units = units
            am2.flush_frames = 1 << 30
            best = None
            for rep in range(2):                                # the first pass warms the context (model upload, lazy buffers)
                t0 = time.perf_counter()
                for n in range(n_sh):
                    am2.multi_embedded_training_1(lab[n], xs[n], False, False, n + 1, n_sh, 0)
                res = am2.flush_workers()
                best = time.perf_counter() - t0
            files = sum(len(f) for _, _, f in os.walk(tmp))
            shim['%d_utterances' % n_sh] = dict(frames_per_s=res['train'][1] / best, ms_per_flush=best * 1e3, utterances=n_sh, frames=res['train'][1],
                                                accumulator_files_written=files // 2)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    out['deferred_batch_shim'] = dict(shim, what='multi_embedded_training_1 with the reference\'s signature, queued; flush_workers() = one estep_batch (model + frames upload, '
                                                  'score, forward-backward, statistics, download) + save_batch_acc (reference-format accumulator files, log domain, one set per '
                                                  'unit); wall clock of calls + flush, incl. every file write')
    return out


if __name__ == '__main__':
    main()
