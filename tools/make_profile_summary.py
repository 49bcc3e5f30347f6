#!/usr/bin/env python3
"""Condense gpurun_out/prof (written by tools/gpu_profile.sh on the GPU box) into the committed summary
profiles/r01_bench_summary.txt + profiles/r01_bench_kernel_stats.csv."""
import csv, glob, collections, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'gpurun_out', 'prof')
out = []
stats = glob.glob(P + '/trace/**/*kernel_stats.csv', recursive=True)[0]
shutil.copy(stats, os.path.join(ROOT, 'profiles', 'r01_bench_kernel_stats.csv'))
out.append('# rocprofv3 summaries, round 1 (final kernels: split-f16 MFMA scoring with LDS-DMA staging, XCD-aware tile order)')
out.append('# command: python3 bench.py --steps 3 --warmup 1 --cpu-baseline 0 --extra 0   (one MI355X, C4 shard)')
out.append('# produced by tools/gpu_profile.sh + tools/make_profile_summary.py; raw CSVs are scratch (gpurun_out/prof)')
out.append('# the f32-input MFMA kernel this replaced: profiles/r01_f32mfma_bench_summary.txt')
out.append('')
out.append('## --kernel-trace --stats (r01_bench_kernel_stats.csv)')
kt = {}
for r in csv.DictReader(open(stats)):
    out.append('%-70s calls=%s avg_ns=%s pct=%s' % (r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage']))
    kt[r['Name'][:70]] = float(r['AverageNs'])
out.append('')
out.append('## --pmc passes (one counter group per pass, no trace domains), per-dispatch averages')
val = {}
for tag in ('pmc_fetch', 'pmc_write', 'pmc_sq', 'pmc_clk'):
    for f in glob.glob('%s/%s/**/*counter_collection.csv' % (P, tag), recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            agg[(r['Kernel_Name'][:60], r['Counter_Name'])][0] += float(r['Counter_Value'])
            agg[(r['Kernel_Name'][:60], r['Counter_Name'])][1] += 1
        for k, v in sorted(agg.items()):
            if 'split16' in k[0] or 'hmm_fb' in k[0] or 'gmm_score_kernel' in k[0]:
                out.append('%-62s %-26s n=%d per-dispatch=%.6g' % (k[0], k[1], v[1], v[0] / v[1]))
            if 'split16' in k[0]:
                val[k[1]] = v[0] / v[1]
ms = [v for k, v in kt.items() if 'split16' in k][0] / 1e6
pairs, M, D = 18432000, 2048, 39
flop = pairs * M * (3 * D + 4)
fetch, write = val['FETCH_SIZE'] * 1024, val['WRITE_SIZE'] * 1024
cyc = val['GRBM_GUI_ACTIVE'] / 8
out += ['', '## derived (gmm_score_split16_kernel<39,2,true>, 18,432,000 (frame,state) pairs x 2048 mixtures per launch)',
        'kernel %.2f ms/launch (rocprofv3 trace pass) -> %.1f TFLOP/s algorithmic (%.4f TFLOP/launch) = %.3f of 838.9 (f16 MFMA peak / 3 split products), %.2f x the f32-input MFMA peak 157.3'
        % (ms, flop / ms / 1e9, flop / 1e12, flop / ms / 1e9 / 838.9, flop / ms / 1e9 / 157.3),
        'executed MFMA work: 15 x v_mfma_f32_32x32x16 per 1024 Gaussians = %.0f TFLOP/s of f16/bf16 MFMA flops' % (pairs * M * 480 / ms / 1e9),
        'GRBM_GUI_ACTIVE %.4g (sum of 8 XCDs) -> %.3g cycles -> clock held %.2f GHz over %.2f ms; SQ_VALU_MFMA_BUSY_CYCLES %.4g / 1024 SIMDs = %.3g cycles -> matrix pipe %.0f %% busy'
        % (val['GRBM_GUI_ACTIVE'], cyc, cyc / ms / 1e6, ms, val['SQ_VALU_MFMA_BUSY_CYCLES'], val['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024, 100 * val['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc),
        'FETCH_SIZE %.4g KiB/launch = %.2f GB; WRITE_SIZE %.4g KiB = %.2f GB' % (val['FETCH_SIZE'], fetch / 1e9, val['WRITE_SIZE'], write / 1e9),
        'algorithmic bytes/launch 2.14 GB (parameters 1.97 GB + frames 48 MB + B 147 MB): traffic/algorithmic = %.2f' % ((fetch + write) / 2.1369e9),
        'HBM %.3f TB/s = %.1f %% of 8 TB/s: compute bound' % ((fetch + write) / ms / 1e9, (fetch + write) / ms / 1e9 / 8 * 100),
        'traffic_bytes for bench.py --traffic-bytes: %d' % int(fetch + write)]
open(os.path.join(ROOT, 'profiles', 'r01_bench_summary.txt'), 'w').write('\n'.join(out) + '\n')
print('\n'.join(out[-10:]))
