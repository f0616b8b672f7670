"""Rewrite one record of profiles/pmc_traffic.json from a fresh rocprofv3 PMC summary (tools/gpu_prof.sh <tag> pmc ->
pmc_summary.txt), stamping it with the hash of the kernel sources it measured (bench.kernel_source_sha), so that bench.py
emits `roofline.traffic` for exactly that build and null for any other.

    python tools/pmc_traffic_update.py profiles/r05_cfg1_f16_pmc_summary.txt 4096x65536_r128_f16_pp pp_kernel
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (only for kernel_source_sha; imports torch)

summary, key, kernel_substr = sys.argv[1:4]
alg = int(sys.argv[4]) if len(sys.argv) > 4 else 590348288
cur, vals = None, {}
for line in open(summary):
    if not line.startswith(' '):
        cur = line.strip()
        continue
    m = re.match(r'\s+(\S+)\s+n=\s*(\d+)\s+mean=(\S+)', line)
    if m and cur and kernel_substr in cur and m.group(1) in ('FETCH_SIZE', 'WRITE_SIZE'):
        vals[m.group(1)] = (float(m.group(3)), int(m.group(2)))
assert 'FETCH_SIZE' in vals and 'WRITE_SIZE' in vals, f'no FETCH_SIZE / WRITE_SIZE rows for a kernel matching {kernel_substr!r}'
path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
d = json.load(open(path))
fam = key.rsplit('_', 1)[1]
d['records'][key] = {
    'kernel': kernel_substr,
    'source': f'{os.path.relpath(summary, ROOT)} (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, '
              f'{vals["FETCH_SIZE"][1]} / {vals["WRITE_SIZE"][1]} launches)',
    'src_sha16': bench.kernel_source_sha(fam),
    'FETCH_SIZE_KB_mean': vals['FETCH_SIZE'][0], 'WRITE_SIZE_KB_mean': vals['WRITE_SIZE'][0],
    'hbm_bytes_per_launch': int(round((2 * vals['FETCH_SIZE'][0] + vals['WRITE_SIZE'][0]) * 1024)),
    'algorithmic_bytes_per_launch': alg}
d['format'] = ("records keyed '<rows>x<cols>_r<rank>_<precision>_<pp|fused|xb>'; src_sha16 = hash of the kernel sources the pass "
               "measured; bench.py emits roofline.traffic only when the run's key is present AND its sources hash the same (null otherwise)")
json.dump(d, open(path, 'w'), indent=1)
print(json.dumps(d['records'][key], indent=1))
