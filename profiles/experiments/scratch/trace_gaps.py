"""kernel durations and start-to-start gaps of energy_direct_d4_kernel from a rocprofv3 kernel trace csv"""
import csv, sys, glob, numpy as np
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
en = sorted([(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'energy_direct_d4' in r['Kernel_Name']])
fin = sorted([(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'cost_finish' in r['Kernel_Name']])
en = np.array(en); print('energy kernels', len(en), 'finish kernels', len(fin))
first_fin = fin[0][0] if fin else 1 << 62
for name, sel in (('before comm', en[:, 1] < first_fin), ('with comm', en[:, 0] > first_fin)):
    e = en[sel]
    if len(e) < 50: continue
    e = e[len(e) // 2:]          # settled half
    dur = e[:, 1] - e[:, 0]; gap = e[1:, 0] - e[:-1, 1]; per = e[1:, 0] - e[:-1, 0]
    print(f'{name}: n={len(e)} dur mean {dur.mean():.0f} ns, gap mean {gap.mean():.0f} ns (median {np.median(gap):.0f}), period mean {per.mean():.0f} median {np.median(per):.0f}')
if fin:
    fa = np.array(fin[len(fin) // 2:]); print('finish kernel dur mean', (fa[:, 1] - fa[:, 0]).mean())
