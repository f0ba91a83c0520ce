"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv + kernel_trace.csv) per kernel name."""
import csv
import glob
import sys
from collections import defaultdict


def main(d, out=None):
    cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
    lines = []
    dur = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    tdur = defaultdict(float)
    for r in csv.DictReader(open(cc[0])):
        k = r['Kernel_Name'][:70]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in cnt[k]:
            cnt[k].add(r['Dispatch_Id'])
            tdur[k] += dur.get(r['Dispatch_Id'], 0.0)
    names = sorted({c for v in agg.values() for c in v})
    lines.append('%-72s %6s %12s ' % ('kernel', 'calls', 'total_us') + ' '.join('%22s' % n for n in names))
    for k in sorted(agg, key=lambda x: -tdur[x]):
        lines.append('%-72s %6d %12.1f ' % (k, len(cnt[k]), tdur[k]) + ' '.join('%22.4g' % agg[k].get(n, 0) for n in names))
    txt = '\n'.join(lines)
    print(txt)
    if out:
        open(out, 'w').write(txt + '\n')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
