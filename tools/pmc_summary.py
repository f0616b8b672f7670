"""Summarise rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch."""
import csv, glob, os, sys, collections
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get('Kernel_Name', '')
            k = k.split('(')[0][:90]
            res[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in sorted(res.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f'    {c:32s} n={len(v):5d} mean={sum(v)/len(v):.6g}')
