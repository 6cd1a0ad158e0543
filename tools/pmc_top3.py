"""gpurun_out/pmc_top3/* -> gpurun_out/r03_top3_traffic.json (copy to profiles/): per-launch HBM-side bytes of the top
conv families next to their algorithmic bytes.  rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE tallies
128-B requests as 64 B (MI355X_MICROARCH.md, HBM section), so it is doubled; dword gathers are not a calibrated access width, so
the read figure is an estimate.  Infinity-Cache hits are counted as traffic."""
import csv, glob, json, os
O = 'gpurun_out/pmc_top3'
CONV = ('igemm_kernel', 'patch3x3_kernel', 'tpatch_kernel', 'wpatch_kernel')
fams = []
for line in open(os.path.join(O, 'families.txt')):
    i, name, mode, B, Cin, H, W, Cout, k, s, p, ups = line.split()
    B, Cin, H, W, Cout, k, s, p, ups = map(int, (B, Cin, H, W, Cout, k, s, p, ups))
    Ho, Wo = ((H << ups) + 2 * p - k) // s + 1, ((W << ups) + 2 * p - k) // s + 1
    x_b, y_b, w_b = 4 * B * Cin * H * W, 4 * B * Cout * Ho * Wo, 4 * Cout * Cin * k * k
    alg = x_b + y_b + w_b          # every operand once (wgrad: reads x and gy, writes w)
    vals = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        f = glob.glob(os.path.join(O, '**', f'f{i}_{c}_counter_collection.csv'), recursive=True)[0]
        per = [float(r['Counter_Value']) for r in csv.DictReader(open(f))
               if r['Counter_Name'] == c and any(kk in r['Kernel_Name'] for kk in CONV)]
        main = sorted(per)[-3:]                      # the timed repetitions of the conv itself (the largest launches of the run)
        vals[c] = sum(main) / len(main) * 1024
    fetch, write = 2.0 * vals['FETCH_SIZE'], vals['WRITE_SIZE']
    fams.append({'name': name, 'mode': mode, 'algorithmic_bytes': alg, 'fetch_bytes_per_launch': fetch, 'fetch_bytes_raw_counter': fetch / 2,
                 'write_bytes_per_launch': write,
                 'traffic_bytes_per_launch': fetch + write, 'traffic_over_algorithmic': (fetch + write) / alg})
out = {'families': fams, 'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/one_conv.py (3 launches of the '
       'conv per run, averaged); KiB -> bytes; FETCH_SIZE doubled (gfx950 tallies 128-B requests as 64 B)'}
json.dump(out, open('gpurun_out/r03_top3_traffic.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
