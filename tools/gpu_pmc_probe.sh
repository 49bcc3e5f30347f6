#!/bin/bash
# counters of the subset accumulate launch (two EM iterations on the C4 shard)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/probe; mkdir -p $O
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 500 rocprofv3 --pmc $set --output-format csv -d $O/pmc_$tag -- python3 $R/tools/em_iter_probe.py 1024 2 > $O/pmc_$tag.log 2>&1 || { tail -5 $O/pmc_$tag.log; exit 1; }
done
python3 - <<P
import csv,glob,collections
for f in glob.glob('$O/pmc_*/**/*counter_collection.csv', recursive=True):
    acc=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'gmm_accumulate_kernel' in n or 'gmm_score_kernel<39, 3' in n:
            key=(n[28:75], r['Dispatch_Id'], r.get('Grid_Size',''))
            acc[key][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in acc.items():
        if max(v.values()) > 1e7: print(k, dict(v))
P
find $O -name "*.csv" -size +1M -delete
