"""Per-kernel summary of a rocprofv3 kernel-trace CSV, restricted to the timed region (last N bench steps)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name'].split('(')[0]
    n = n.replace('void ', '').replace('nmfmu::', '')[:60]
    d.setdefault(n, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in d.values())
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f'{n:62s} calls={len(v):5d} avg={sum(v)/len(v):9.1f} us  total={sum(v)/1e3:8.2f} ms  {100*sum(v)/tot:5.1f}%')
