"""Summarise gpurun_out/pmc_<tag>/{FETCH_SIZE,WRITE_SIZE}/r_counter_collection.csv per kernel family and training step:
    python tools/pmc_traffic_parse.py <tag> <steps_profiled> <out.json> <out.txt>"""
import csv, sys, re, json, collections
tag, steps, out_json, out_txt = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
def fam(n):
    m = re.search(r'(k_[a-z0-9_]+)', n)
    if m:
        f = m.group(1)
        if f in ('k_conv_patch', 'k_conv_pers'):
            a = re.search(r'k_conv_p\w+<(\w+), (\w+)', n)
            return f + ('(forward)' if a.group(1) == 'true' else '(data-gradient)')
        return f
    return 'at::native / other'
tot = {c: collections.defaultdict(float) for c in ('FETCH_SIZE', 'WRITE_SIZE')}
cnt = collections.defaultdict(int)
for c in tot:
    for r in csv.DictReader(open(f'gpurun_out/pmc_{tag}/{c}/r_counter_collection.csv')):
        if r['Counter_Name'] != c: continue
        tot[c][fam(r['Kernel_Name'])] += float(r['Counter_Value']) * 1024.0      # counters are KiB
        if c == 'FETCH_SIZE': cnt[fam(r['Kernel_Name'])] += 1
rows = sorted(cnt, key=lambda k: -(2 * tot['FETCH_SIZE'][k] + tot['WRITE_SIZE'][k]))
res = {'steps_profiled': steps, 'note': 'bytes per training step; read bytes = 2 x FETCH_SIZE (gfx950 correction for wide streaming reads, MI355X_MICROARCH.md)', 'families': {}}
lines = ['# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), bench.py --steps 3 --warmup 1',
         '# per training step (B=8); read GB = 2 x FETCH_SIZE (gfx950 correction), write GB = WRITE_SIZE',
         '%-32s %9s %10s %10s %14s' % ('kernel family', 'launches', 'read GB', 'write GB', 'MB / launch')]
tr = tw = 0.0
for k in rows:
    n = cnt[k] / steps; rd = 2 * tot['FETCH_SIZE'][k] / steps; wr = tot['WRITE_SIZE'][k] / steps
    tr += rd; tw += wr
    res['families'][k] = {'launches_per_step': n, 'read_bytes_per_step': rd, 'write_bytes_per_step': wr,
                          'bytes_per_launch': (rd + wr) / max(n, 1e-9)}
    lines.append('%-32s %9.1f %10.3f %10.3f %14.2f' % (k, n, rd / 1e9, wr / 1e9, (rd + wr) / max(n, 1e-9) / 1e6))
lines.append('total per step: read %.2f GB + write %.2f GB = %.2f GB; algorithmic (ideal fusion, fp32 storage): 7.62 GB' % (tr / 1e9, tw / 1e9, (tr + tw) / 1e9))
res['total_read_bytes_per_step'] = tr; res['total_write_bytes_per_step'] = tw
json.dump(res, open(out_json, 'w'), indent=1)
open(out_txt, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
