import csv, collections, sys
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    n=r['Kernel_Name']
    if 'k_flash' not in n: continue
    key=n.split('(')[0][:40]+' grid='+r['Grid_Size']
    agg[key][r['Counter_Name']]+=float(r['Counter_Value']); 
    if r['Counter_Name']==sys.argv[2]: cnt[key]+=1
for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get(sys.argv[2],0))[:8]:
    print(k, 'launches',cnt[k], {a: '%.4g'%(b/max(cnt[k],1)) for a,b in v.items()})
