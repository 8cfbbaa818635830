#!/usr/bin/env python3
"""Turn a gpurun_out/prof_<tag>/ directory (written by tools/prof.sh on the GPU box) into the committed summaries
under profiles/: <tag>_kernel_stats.csv, <tag>_timed_region.json, <tag>_pmc.json and the traffic.json table bench.py
reads (whole-step HBM bytes = sum over the kernels of one step of FETCH_SIZE x 2 + WRITE_SIZE).

usage: tools/collect_profiles.py <tag> <D> <B> <solver> <store_env 0|1> <rotate> [bytes per evaluation]
(solver 'overlap': the time-evolution overlap workload, algorithmic bytes 32 D^2 + 16 per candidate)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, D, B, solver, store_env, rotate = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
src = os.path.join(ROOT, 'gpurun_out', f'prof_{tag}')
short = lambda k: k.split('(')[0].replace('void ', '')
stats = glob.glob(os.path.join(src, 'trace', '**', '*kernel_stats.csv'), recursive=True)[0]
shutil.copy(stats, os.path.join(ROOT, 'profiles', f'{tag}_kernel_stats.csv'))
# per-kernel averages over the TIMED region of the traced run (the last `steps` launches of each kernel), next to
# rocprofv3's own all-launch statistics
trace = glob.glob(os.path.join(src, 'trace', '**', '*kernel_trace.csv'), recursive=True)
steps = 200
bench_line = None
for line in open(os.path.join(src, 'bench_trace.log')):
    if line.startswith('{'):
        bench_line = json.loads(line)
        steps = bench_line['steps']
if trace:
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(trace[0])):
        per[short(r['Kernel_Name'])].append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    timed = {}
    for k, v in per.items():
        if 'qmps' in k and 'probe' not in k:
            v.sort()
            last = [d for _, d in v[-steps:]]
            timed[k] = {'launches_total': len(v), 'timed_region_launches': len(last),
                        'timed_region_average_ns': sum(last) / len(last), 'all_launch_average_ns': sum(d for _, d in v) / len(v)}
    json.dump({'command': 'rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extras ' + os.environ.get('BENCH_EXTRA', ''), 'steps': steps,
               'bench_line_of_the_traced_run': bench_line, 'kernels': timed},
              open(os.path.join(ROOT, 'profiles', f'{tag}_timed_region.json'), 'w'), indent=1)
    print(json.dumps(timed, indent=1))
pmc = {}
for f in glob.glob(os.path.join(src, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        if 'qmps' in k and 'probe' not in k:
            pmc.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in d.items()})
            pmc[k]['dispatches_sampled'] = len(next(iter(d.values())))
step_bytes, detail = 0.0, {}
for k, d in pmc.items():
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        b = (2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024
        detail[k] = {'bytes': b, 'fetch_kb_x2': 2 * d['FETCH_SIZE'], 'write_kb': d['WRITE_SIZE']}
        step_bytes += b
pmc['_notes'] = {'cmd': f'bench.py --steps 5 --warmup 1 --D {D} --batch {B} --solver {solver} --rotate {rotate}'
                        f'{" --store-env" if store_env else ""} under rocprofv3 --pmc (separate passes per counter group, tools/prof.sh)',
                 'units': 'means per dispatch; FETCH_SIZE / WRITE_SIZE in KB; gfx950 FETCH_SIZE counts 1/2 of wide '
                          'coalesced reads -> x2 (MI355X_MICROARCH.md, HBM section)'}
json.dump(pmc, open(os.path.join(ROOT, 'profiles', f'{tag}_pmc.json'), 'w'), indent=1)
tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
table = json.load(open(tpath)) if os.path.exists(tpath) else {}
if detail:
    dom = max(detail, key=lambda k: detail[k]['bytes'])
    per_eval = int(sys.argv[7]) if len(sys.argv) > 7 else (32 * D * D + (16 if solver == 'overlap' else 8))
    alg = B * per_eval
    table[f'D={D}|B={B}|solver={solver}|store_env={store_env}|rotate={rotate}'] = {
        'bytes': detail[dom]['bytes'], 'unit': 'B per launch of the dominant kernel', 'kernel': dom,
        'step_bytes': step_bytes, 'algorithmic_bytes': alg, 'dominant_over_algorithmic': detail[dom]['bytes'] / alg,
        'step_over_algorithmic': step_bytes / alg, 'kernels_of_a_step': detail, 'source': f'profiles/{tag}_pmc.json'}
json.dump(table, open(tpath, 'w'), indent=1)
print(open(os.path.join(ROOT, 'profiles', f'{tag}_kernel_stats.csv')).read())
print(json.dumps(table, indent=1))
