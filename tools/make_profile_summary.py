#!/usr/bin/env python3
"""Condense the rocprofv3 passes of tools/gpu_profile.sh into <dir>/<round>_bench_summary.txt, <round>_bench_kernel_stats.csv,
<round>_accumulate_summary.txt, <round>_accumulate_kernel_stats.csv (copied to profiles/ afterwards).
usage: make_profile_summary.py <dir with the passes> <round tag>"""
import collections
import csv
import glob
import os
import shutil
import sys

P, RND = sys.argv[1], sys.argv[2]


def kernel_stats(tag, dst):
    f = glob.glob('%s/%s/**/*kernel_stats.csv' % (P, tag), recursive=True)
    if not f:
        return {}, ['(trace pass %s missing)' % tag]
    shutil.copy(f[0], os.path.join(P, dst))
    out, kt = [], {}
    for r in csv.DictReader(open(f[0])):
        out.append('%-72s calls=%s avg_ns=%s pct=%s' % (r['Name'][:72], r['Calls'], r['AverageNs'], r['Percentage']))
        kt[r['Name']] = float(r['AverageNs'])
    return kt, out


def counters(tags, keep):
    out, val = [], collections.defaultdict(dict)
    for tag in tags:
        for f in glob.glob('%s/%s/**/*counter_collection.csv' % (P, tag), recursive=True):
            agg = collections.defaultdict(lambda: [0.0, 0])
            for r in csv.DictReader(open(f)):
                k = (r['Kernel_Name'], r['Counter_Name'])
                agg[k][0] += float(r['Counter_Value'])
                agg[k][1] += 1
            for k, v in sorted(agg.items()):
                if any(s in k[0] for s in keep):
                    out.append('%-62s %-26s n=%d per-dispatch=%.6g' % (k[0][:62], k[1], v[1], v[0] / v[1]))
                    val[[s for s in keep if s in k[0]][0]][k[1]] = v[0] / v[1]
    return out, val


def pick(kt, name):
    return [v for k, v in kt.items() if name in k][0] / 1e6


# ---------------------------------------------------------------- bench.py (score + forward-backward)
out = ['# rocprofv3 summaries, %s: python3 bench.py --steps 20 --warmup 2 --cpu-baseline 0 --extra 0 --sustain 0 (one MI355X, C4 shard)' % RND,
       '# produced by tools/gpu_profile.sh + tools/make_profile_summary.py; one --kernel-trace --stats pass and separate --pmc passes',
       '', '## --kernel-trace --stats (%s_bench_kernel_stats.csv)' % RND]
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _root)
import bench as _bench                       # (no GPU call at import)
from poccala_amd import synth as _synth
# identity of the kernel CODE the passes measured (its body and compile flags, not the launch side of the file): bench.py withholds
# `traffic` when the kernel has changed since
out.insert(2, 'kernel_code_sha16 gmm_score_split16_kernel %s' % _bench.scoring_kernel_sha16())
kt, lines = kernel_stats('bench_trace', '%s_bench_kernel_stats.csv' % RND)
out += lines + ['', '## --pmc passes (<= 4 counters per pass, no trace domains), per-dispatch averages']
lines, val = counters(['bench_fetch', 'bench_write', 'bench_clk', 'bench_sq1', 'bench_sq2'], ['gmm_score_split16_kernel', 'hmm_fbl_kernel', 'hmm_postl_kernel', 'hmm_emis_pack_kernel'])
out += lines
try:
    v = val['gmm_score_split16_kernel']
    ms = pick(kt, 'gmm_score_split16_kernel')
    # the (frame, state) pairs a launch SCORES: a label that names a unit twice is scored once and copied (bench.py: scored_pairs), mean over
    # the two resident batches the profiled command alternates between (seeds as in bench.py main(), rank 0)
    _c = _synth.CONFIGS['C4shard']
    _labels = _synth.make_labels(_c['U'] * 2, _c['L'], _c['units'], seed=2)
    pairs = int(round(sum(3 * len(set(l.tolist())) * _c['T'] for l in _labels) / 2))
    M, D = _c['M'], _c['D']
    flop = pairs * M * (3 * D + 4)
    fetch, write = v['FETCH_SIZE'] * 1024, v['WRITE_SIZE'] * 1024
    cyc = v['GRBM_GUI_ACTIVE'] / 8
    out += ['', '## derived (gmm_score_split16_kernel<39,2>, %d scored (frame,state) pairs x 2048 mixtures per launch; 18,432,000 label pairs, repeats are copied)' % pairs,
            'kernel %.2f ms/launch (trace pass) -> %.1f TFLOP/s algorithmic (%.4f TFLOP/launch) = %.3f of the dense f16 MFMA peak 2516.6 = %.3f of 838.9 (that peak / 3 split products), %.2f x the f32-input MFMA peak 157.3'
            % (ms, flop / ms / 1e9, flop / 1e12, flop / ms / 1e9 / 2516.6, flop / ms / 1e9 / 838.9, flop / ms / 1e9 / 157.3),
            'executed MFMA work: 15 x v_mfma_f32_32x32x16 per 1024 Gaussians = %.0f TFLOP/s of f16 MFMA flops = %.2f of the 2516.6 dense peak' % (pairs * M * 480 / ms / 1e9, pairs * M * 480 / ms / 1e9 / 2516.6),
            'GRBM_GUI_ACTIVE %.4g (sum of 8 XCDs) -> %.3g cycles -> clock held %.2f GHz over %.2f ms; SQ_VALU_MFMA_BUSY_CYCLES %.4g / 1024 SIMDs = %.3g cycles -> matrix pipe %.0f %% busy'
            % (v['GRBM_GUI_ACTIVE'], cyc, cyc / ms / 1e6, ms, v['SQ_VALU_MFMA_BUSY_CYCLES'], v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024, 100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc),
            'FETCH_SIZE %.4g KiB/launch = %.2f GB raw; WRITE_SIZE %.4g KiB = %.2f GB' % (v['FETCH_SIZE'], fetch / 1e9, v['WRITE_SIZE'], write / 1e9),
            'gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies wide streaming reads (16 B/lane, global_load_lds included: the parameter '
            'stream) at half their bytes -> traffic = 2 x FETCH_SIZE + WRITE_SIZE = %.2f GB per launch (an upper bound: the 156-B frame rows are narrow reads)' % ((2 * fetch + write) / 1e9),
            'algorithmic bytes/launch 2.14 GB (parameters 1.97 GB + frames 48 MB + B 147 MB): corrected traffic / algorithmic = %.2f' % ((2 * fetch + write) / 2.1369e9),
            'HBM %.3f TB/s = %.1f %% of 8 TB/s: compute bound' % ((2 * fetch + write) / ms / 1e9, (2 * fetch + write) / ms / 1e9 / 8 * 100),
            'traffic_bytes for bench.py: %d' % int(2 * fetch + write)]
except (KeyError, IndexError) as e:
    out.append('(derived block incomplete: %r)' % (e,))
open(os.path.join(P, '%s_bench_summary.txt' % RND), 'w').write('\n'.join(out) + '\n')

# ---------------------------------------------------------------- accumulate pass (tools/acc_bench.py)
ACC_PASSES = int(os.environ.get('ACC_PASSES', '4'))
out = ['# rocprofv3 summaries, %s: python3 tools/acc_bench.py (one MI355X, C4 shard, flat posteriors: %d accumulate passes)' % (RND, ACC_PASSES),
       '', '## --kernel-trace --stats (%s_accumulate_kernel_stats.csv)' % RND]
kt, lines = kernel_stats('acc_trace', '%s_accumulate_kernel_stats.csv' % RND)
out += lines + ['', '## --pmc passes (<= 4 counters per pass), per-dispatch averages']
lines, val = counters(['acc_fetch', 'acc_write', 'acc_clk', 'acc_sq1', 'acc_sq2', 'acc_lds', 'acc_tcc'], ['acc16_consumer_kernel', 'acc16_producer_kernel'])
out += lines
try:
    v = val['acc16_consumer_kernel']
    ms = pick(kt, 'acc16_consumer_kernel')
    calls = [r for r in csv.DictReader(open(os.path.join(P, '%s_accumulate_kernel_stats.csv' % RND))) if 'acc16_consumer' in r['Name']][0]
    n = int(calls['Calls'])
    cyc = v['GRBM_GUI_ACTIVE'] / 8
    mf = v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024
    wc = v['SQ_WAVE_CYCLES']
    out += ['', '## derived (acc16_consumer_kernel<39>, per dispatch = one state group; %d dispatches in %d passes)' % (n, ACC_PASSES),
            'consumer %.2f ms/dispatch, producer %.2f ms/dispatch; per pass: %d groups' % (ms, pick(kt, 'acc16_producer_kernel'), n // ACC_PASSES),
            'GRBM_GUI_ACTIVE %.4g (sum of 8 XCDs) -> clock held %.2f GHz; SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs = %.3g cycles -> matrix pipe %.0f %% busy'
            % (v['GRBM_GUI_ACTIVE'], cyc / ms / 1e6, mf, 100 * mf / cyc),
            'wave time: SQ_WAIT_ANY %.0f %% (s_waitcnt / barrier), SQ_WAIT_INST_ANY %.0f %% (issue stalls: pipe, dependencies), SQ_ACTIVE_INST_ANY %.0f %% of SQ_WAVE_CYCLES'
            % (100 * v['SQ_WAIT_ANY'] / wc, 100 * v['SQ_WAIT_INST_ANY'] / wc, 100 * v['SQ_ACTIVE_INST_ANY'] / wc),
            'LDS: SQ_LDS_IDX_ACTIVE / 256 CUs = %.3g cycles = %.0f %% of the dispatch, bank conflicts %.3g' % (v['SQ_LDS_IDX_ACTIVE'] / 256, 100 * v['SQ_LDS_IDX_ACTIVE'] / 256 / cyc, v['SQ_LDS_BANK_CONFLICT']),
            'L2: hit rate %.0f %% (TCC_HIT / (HIT + MISS)); FETCH_SIZE %.2f GB raw (x 2 for 16-B-per-lane streaming reads = %.2f GB), WRITE_SIZE %.2f GB per dispatch'
            % (100 * v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum']), v['FETCH_SIZE'] * 1024 / 1e9, 2 * v['FETCH_SIZE'] * 1024 / 1e9, v['WRITE_SIZE'] * 1024 / 1e9)]
except (KeyError, IndexError, ZeroDivisionError) as e:
    out.append('(derived block incomplete: %r)' % (e,))
open(os.path.join(P, '%s_accumulate_summary.txt' % RND), 'w').write('\n'.join(out) + '\n')

# ---------------------------------------------------------------- forward-backward alone (tools/fb_bench.py: 62-state sentence HMMs x 300 frames)
out = ['# rocprofv3 summaries, %s: python3 tools/fb_bench.py U (one MI355X; U sentence HMMs of 62 states x 300 frames, random emissions, nothing beside them;' % RND,
       '# 60 + 20 launches with a free pi (3 passes) and as many with a locked one: the scaled linear-domain route of csrc/hmm_fb_linear.inc)']
for U in (128, 1024):
    kt, lines = kernel_stats('fb%d_trace' % U, '%s_fb%d_kernel_stats.csv' % (RND, U))
    out += ['', '## U = %d: --kernel-trace --stats' % U] + [l for l in lines if 'hmm_' in l]
    lines, val = counters(['fb%d_fetch' % U, 'fb%d_write' % U, 'fb%d_clk' % U, 'fb%d_sq1' % U, 'fb%d_sq2' % U], ['hmm_fbl_kernel', 'hmm_postl_kernel', 'hmm_emis_pack_kernel'])
    out += ['## U = %d: --pmc passes, per-dispatch averages' % U] + lines
    try:
        steps = 299 * 3.5                       # chain steps per launch, averaged over the free-pi (4 walks: beta + 3 alpha) and locked-pi (2 walks) halves
        for name in ('hmm_emis_pack_kernel', 'hmm_fbl_kernel', 'hmm_postl_kernel'):
            v = val[name]
            ms = pick(kt, name)
            cyc = v['GRBM_GUI_ACTIVE'] / 8
            wc = v['SQ_WAVE_CYCLES']
            gb = (v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024 / 1e9
            out.append('derived %-22s %.3f ms/launch, FETCH+WRITE %.3f GB raw = %.2f TB/s; wave time: waiting (s_waitcnt / barrier) %.0f %%, issue stalls %.0f %%, issuing %.0f %%; %.3g VALU + %.3g SALU instructions per wave-launch'
                       % (name, ms, gb, gb / ms, 100 * v['SQ_WAIT_ANY'] / wc, 100 * v['SQ_WAIT_INST_ANY'] / wc, 100 * v['SQ_ACTIVE_INST_ANY'] / wc,
                          v['SQ_INSTS_VALU'] / (U * (2 if 'fbl' in name else 8)), v['SQ_INSTS_SALU'] / (U * (2 if 'fbl' in name else 8))))
    except (KeyError, IndexError, ZeroDivisionError) as e:
        out.append('(derived block incomplete: %r)' % (e,))
open(os.path.join(P, '%s_fb_summary.txt' % RND), 'w').write('\n'.join(out) + '\n')

# ---------------------------------------------------------------- C5 shard: all-state scoring + token passing (tools/c5_decode_bench.py)
out = ['# rocprofv3 summaries, %s: python3 tools/c5_decode_bench.py 417 4096 20000 3 8192 (one MI355X: BASELINE config 5 per-GPU shard, 417 x 300 frames,' % RND,
       '# 549 states x 4096 mixtures, 20 k-word tree, at most 8192 live tokens per utterance; resident runs + a streamed run in 3 chunks)',
       '', '## --kernel-trace --stats (%s_decode_kernel_stats.csv)' % RND]
kt, lines = kernel_stats('dec_trace', '%s_decode_kernel_stats.csv' % RND)
out += lines + ['', '## --pmc passes (<= 4 counters per pass), per-dispatch averages (all launches: full shard and chunks)']
lines, val = counters(['dec_fetch', 'dec_write', 'dec_clk', 'dec_sq1'], ['hmm_decode_']) if True else None
out += lines
try:
    v = val['hmm_decode_']
    ms = pick(kt, 'hmm_decode_')
    cyc = v['GRBM_GUI_ACTIVE'] / 8
    wc = v['SQ_WAVE_CYCLES']
    out += ['', '## derived (hmm_decode_lr_kernel / hmm_decode_kernel, averages over the launches above)',
            'FETCH_SIZE %.2f GB raw + WRITE_SIZE %.2f GB per launch of %.1f ms average -> %.2f TB/s raw (the token arrays are read with 8- and 4-byte lanes: no x 2)'
            % (v['FETCH_SIZE'] * 1024 / 1e9, v['WRITE_SIZE'] * 1024 / 1e9, ms, (v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024 / ms / 1e9),
            'clock held %.2f GHz; wave time: SQ_WAIT_ANY %.0f %%, SQ_WAIT_INST_ANY %.0f %% of SQ_WAVE_CYCLES' % (cyc / ms / 1e6, 100 * v['SQ_WAIT_ANY'] / wc, 100 * v['SQ_WAIT_INST_ANY'] / wc)]
except (KeyError, IndexError, ZeroDivisionError) as e:
    out.append('(derived block incomplete: %r)' % (e,))
open(os.path.join(P, '%s_decode_summary.txt' % RND), 'w').write('\n'.join(out) + '\n')

# ---------------------------------------------------------------- the coarse pass (tools/coarse_time_probe.py time: the C4 shard's model after two EM iterations)
out = ['# rocprofv3 summaries, %s: python3 tools/coarse_time_probe.py time (one MI355X; C4 shard, the model two EM iterations at the variance floor 1e-6' % RND,
       '# leave: 73 % of the mixtures off the matrix pipe; four scoring calls = main kernel over the on-pipe 27 % + the coarse pass over the rest)',
       '', '## --kernel-trace --stats (%s_coarse_kernel_stats.csv)' % RND]
kt, lines = kernel_stats('coarse_trace', '%s_coarse_kernel_stats.csv' % RND)
out += lines + ['', '## --pmc passes (<= 4 counters per pass), per-dispatch averages']
lines, val = counters(['coarse_fetch', 'coarse_write', 'coarse_clk', 'coarse_sq1', 'coarse_sq2', 'coarse_lds'], ['gmm_score_coarse_kernel', 'gmm_score_split16_kernel'])
out += lines
try:
    for name in ('gmm_score_coarse_kernel', 'gmm_score_split16_kernel'):
        v = val[name]
        ms = pick(kt, name)
        cyc = v['GRBM_GUI_ACTIVE'] / 8
        wc = v['SQ_WAVE_CYCLES']
        out += ['', '## derived, %s (%.2f ms per launch)' % (name, ms),
                'FETCH_SIZE %.3f GB raw (x 2: the layouts are read with 16-byte lanes) + WRITE_SIZE %.3f GB per launch' % (v['FETCH_SIZE'] * 1024 / 1e9, v['WRITE_SIZE'] * 1024 / 1e9),
                'clock held %.2f GHz; matrix pipe busy %.0f %% of the SIMD cycles (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs))'
                % (cyc / ms / 1e6, 100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024)),
                'wave time: SQ_WAIT_ANY %.0f %%, SQ_WAIT_INST_ANY %.0f %%, SQ_WAIT_INST_LDS %.0f %% of SQ_WAVE_CYCLES; LDS bank conflicts %.1f %% of SQ_LDS_IDX_ACTIVE'
                % (100 * v['SQ_WAIT_ANY'] / wc, 100 * v['SQ_WAIT_INST_ANY'] / wc, 100 * v.get('SQ_WAIT_INST_LDS', 0) / wc,
                   100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, v.get('SQ_LDS_IDX_ACTIVE', 1.0)))]
except (KeyError, IndexError, ZeroDivisionError) as e:
    out.append('(derived block incomplete: %r)' % (e,))
if kt:
    open(os.path.join(P, '%s_coarse_summary.txt' % RND), 'w').write('\n'.join(out) + '\n')
