import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0,0.0,0.0])
for r in rows:
    k=(r['kind'], r['label'])
    agg[k][0]+=1; agg[k][1]+=float(r['ms']); agg[k][2]+=float(r['gflop'])
tot = sum(v[1] for v in agg.values())
print('total ms', round(tot,3), 'ops', len(rows))
n = int(sys.argv[2]) if len(sys.argv)>2 else 40
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:n]:
    tf = v[2]/v[1] if v[1] else 0
    print(f"{v[1]:7.3f} ms  n={v[0]:3d}  {tf:7.1f} TF/s  avg {1e3*v[1]/v[0]:7.1f} us  {k[0]:22s} {k[1]}")
