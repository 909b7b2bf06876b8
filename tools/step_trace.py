"""Per-kernel durations of ONE training step from a rocprofv3 --kernel-trace csv (the step between the last two Adam
launches but one).  Usage: python tools/step_trace.py trace.csv [--all]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
agg = collections.OrderedDict()
for r in rows[a + 1:b + 1]:
    n = r['Kernel_Name'].replace('nerf::', '').replace('void ', '').split('(')[0][:44]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if '--all' in sys.argv:
        print('%-46s %8.1f us grid %s' % (n, d, r['Grid_Size_X']))
    e = agg.setdefault(n, [0, 0.0]); e[0] += 1; e[1] += d
tot = sum(v[1] for v in agg.values())
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-46s x%-3d %8.1f us  %5.1f%%' % (n, c, d, 100 * d / tot))
print('kernel time %.1f us, span %.1f us' % (tot, (int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3))
