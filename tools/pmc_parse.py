import csv, sys, collections, glob, re
tag = sys.argv[1]; pat = sys.argv[2]
agg = collections.defaultdict(list)
for f in sorted(glob.glob(f'gpurun_out/pmc_{tag}/p*/r_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if re.search(pat, r['Kernel_Name']):
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: sum(v)/len(v) for k, v in agg.items()}
for k in sorted(res): print(f"{k:32s} {res[k]:16.0f}")
w = res.get('SQ_WAVES', 1)
print("per wave: VALU %.0f MFMA %.0f LDS %.0f SALU %.0f VMEM_RD %.0f VMEM_WR %.0f CVT %.0f" % tuple(res.get(k,0)/w for k in
      ['SQ_INSTS_VALU','SQ_INSTS_MFMA','SQ_INSTS_LDS','SQ_INSTS_SALU','SQ_INSTS_VMEM_RD','SQ_INSTS_VMEM_WR','SQ_INSTS_VALU_CVT']))
bc = res.get('SQ_BUSY_CYCLES', 1)
for k in ['SQ_VALU_MFMA_BUSY_CYCLES','SQ_ACTIVE_INST_VALU','SQ_ACTIVE_INST_LDS','SQ_WAIT_INST_LDS','SQ_WAIT_INST_ANY','SQ_WAIT_ANY','SQ_LDS_BANK_CONFLICT','SQ_LDS_IDX_ACTIVE','SQ_ACTIVE_INST_ANY','SQ_ACTIVE_INST_VMEM','SQ_WAVE_CYCLES','SQ_INST_CYCLES_VMEM_RD','SQ_LDS_ADDR_CONFLICT','SQ_LDS_UNALIGNED_STALL']:
    if k in res: print(f"{k:28s} / SQ_BUSY_CYCLES = {res[k]/bc:8.3f}")
