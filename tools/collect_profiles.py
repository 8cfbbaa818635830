#!/usr/bin/env python3
"""Turn a gpurun_out/prof_<tag>/ directory (written by tools/prof.sh on the GPU box) into the committed
summaries under profiles/: <tag>_kernel_stats.csv, <tag>_pmc.json and the traffic.json table bench.py reads."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, D, B, solver, handoff = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
src = os.path.join(ROOT, 'gpurun_out', f'prof_{tag}')
stats = glob.glob(os.path.join(src, 'trace', '**', '*kernel_stats.csv'), recursive=True)[0]
shutil.copy(stats, os.path.join(ROOT, 'profiles', f'{tag}_kernel_stats.csv'))
# per-kernel averages over the TIMED region of the traced run (the last `steps` launches of each kernel; the warm-up
# launches run while the clocks are still ramping), next to rocprofv3's own all-launch statistics
trace = glob.glob(os.path.join(src, 'trace', '**', '*kernel_trace.csv'), recursive=True)
if trace:
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(trace[0])):
        per[r['Kernel_Name'].split('(')[0].replace('void ', '')].append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    steps = 200
    for line in open(os.path.join(src, 'bench_trace.log')):
        if line.startswith('{'):
            steps = json.loads(line)['steps']
    timed = {}
    for k, v in per.items():
        if 'qmps' in k:
            v.sort()
            last = [d for _, d in v[-steps:]]
            timed[k] = {'launches_total': len(v), 'timed_region_launches': len(last),
                        'timed_region_average_ns': sum(last) / len(last), 'all_launch_average_ns': sum(d for _, d in v) / len(v)}
    json.dump({'command': 'rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extras', 'steps': steps,
               'kernels': timed}, open(os.path.join(ROOT, 'profiles', f'{tag}_timed_region.json'), 'w'), indent=1)
    print(json.dumps(timed, indent=1))
pmc = {}
for f in glob.glob(os.path.join(src, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        if 'qmps' in k:
            pmc.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in d.items()})
pmc['_notes'] = {'cmd': f'bench.py --steps 5 --warmup 1 --D {D} --batch {B} --solver {solver} --handoff {handoff} under '
                        'rocprofv3 --pmc (separate passes per counter group, tools/prof.sh)',
                 'units': 'means per dispatch; FETCH_SIZE / WRITE_SIZE in KB; gfx950 FETCH_SIZE counts 1/2 of wide '
                          'coalesced reads -> x2 (MI355X_MICROARCH.md, HBM section)'}
json.dump(pmc, open(os.path.join(ROOT, 'profiles', f'{tag}_pmc.json'), 'w'), indent=1)
tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
table = json.load(open(tpath)) if os.path.exists(tpath) else {}
names = {'qmps::env_square_d4_kernel': 'env_square_d4_kernel',
         'qmps::energy_lane_kernel<4, true>': 'energy_lane_kernel<4,true>'}
for k, d in pmc.items():
    if k in names and 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        key = f'{names[k]}|D={D}|B={B}|solver={solver}|handoff={handoff}'
        table[key] = {'bytes': (2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024, 'fetch_kb_x2': 2 * d['FETCH_SIZE'],
                      'write_kb': d['WRITE_SIZE'], 'source': f'profiles/{tag}_pmc.json'}
json.dump(table, open(tpath, 'w'), indent=1)
print(open(os.path.join(ROOT, 'profiles', f'{tag}_kernel_stats.csv')).read())
print(json.dumps(table, indent=1))
