"""Summarise tools/pmc.sh passes (gpurun_out/pmc_<tag>/p*_counter_collection.csv) into one line per conv kernel:
MFMA busy share = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs * kernel cycles), kernel cycles from SQ_BUSY_CYCLES / 32 SEs."""
import csv, glob, collections, sys, json
tag = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f'gpurun_out/pmc_{tag}/p*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'igemm_kernel' in n or 'patch3x3' in n or 'rgb3x3' in n:
            short = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(IgemmParams')[0]
            vals[short][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, c in vals.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = m.get('SQ_BUSY_CYCLES', 0) / 32.0
    d = {'launches_sampled': len(next(iter(c.values()))), 'kernel_cycles': cyc}
    if cyc and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
        d['mfma_busy_share'] = m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc)
    for n in ('SQ_INSTS_MFMA', 'SQ_INSTS_VALU', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'SQ_LDS_BANK_CONFLICT', 'SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES'):
        if n in m:
            d[n] = m[n]
    out[k] = d
print(json.dumps(out, indent=1))
json.dump(out, open(f'profiles/r01_pmc_{tag}.json', 'w'), indent=1)
