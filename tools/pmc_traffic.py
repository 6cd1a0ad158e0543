"""Summarise gpurun_out/pmc_traffic/{FETCH_SIZE,WRITE_SIZE}_counter_collection.csv into profiles/<tag>_igemm_traffic.json:
HBM-side bytes per conv-kernel launch. rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE tallies 128-B requests
as 64 B (MI355X_MICROARCH.md, HBM section), so it is doubled. Dword gathers are not a calibrated access width, so the
number is an estimate of fabric-side traffic (Infinity-Cache hits included), not an exact byte count."""
import csv, json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
out = {}
CONV = ('igemm_kernel', 'patch3x3_kernel', 'rgb3x3_kernel', 'gather_small_m_kernel')     # every conv launch, as bench.py's roofline counts them
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    tot = 0.0; n = 0
    for r in csv.DictReader(open(f'gpurun_out/pmc_traffic/{c}_counter_collection.csv')):
        if any(k in r['Kernel_Name'] for k in CONV) and r['Counter_Name'] == c:
            tot += float(r['Counter_Value']); n += 1
    out[c] = {'kib_total': tot, 'launches': n}
n = out['FETCH_SIZE']['launches']
fetch = 2.0 * out['FETCH_SIZE']['kib_total'] * 1024 / n
write = out['WRITE_SIZE']['kib_total'] * 1024 / max(out['WRITE_SIZE']['launches'], 1)
res = {'kernel': 'igemm_kernel + patch3x3_kernel + rgb3x3_kernel + gather_small_m_kernel', 'launches': n, 'fetch_bytes_per_launch': fetch, 'write_bytes_per_launch': write,
       'traffic_bytes_per_launch': fetch + write,
       'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 1 --warmup 1; KiB -> bytes, '
                 'FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B)'}
json.dump(res, open(f'profiles/{tag}_igemm_traffic.json', 'w'), indent=1)
print(res)
