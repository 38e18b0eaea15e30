#!/bin/bash
# instruction counts per phase of k_conv_pers (u02 forward, 32->72 3x3): one PMC pass per MD_DBG ablation
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES"
for dbg in 0 2 8 16 32 64 90; do
  MD_DBG=$dbg timeout -k 10 200 rocprofv3 --pmc $P1 --output-format csv -d $O/p$dbg -o r -- python3 $R/tools/pmc_layer.py c1s fwd > $O/log$dbg.txt 2>&1 || exit 1
  python3 - <<PY | tee -a $O/abl_counts.txt
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('$O/p$dbg/**/r_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_conv_pers' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: sum(v)/len(v) for k, v in agg.items()}
w = res.get('SQ_WAVES', 1)
print('dbg=$dbg per wave:', ' '.join('%s %.0f' % (k.replace('SQ_INSTS_',''), res.get(k,0)/w) for k in ['SQ_INSTS_VALU','SQ_INSTS_MFMA','SQ_INSTS_LDS','SQ_INSTS_SALU','SQ_INSTS_VMEM_RD','SQ_INSTS_VMEM_WR']), 'wave_cycles/wave %.0f' % (res.get('SQ_WAVE_CYCLES',0)/w))
PY
  rm -rf $O/p$dbg
done
