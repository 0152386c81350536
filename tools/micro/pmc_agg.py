# usage: pmc_agg.py <kernel substring> csv...   -> mean counter value per launch for that kernel
import csv, sys, collections
sub=sys.argv[1]; agg=collections.defaultdict(float); cnt=collections.Counter()
for p in sys.argv[2:]:
    for r in csv.DictReader(open(p)):
        if sub in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value']); cnt[r['Counter_Name']]+=1
for k in sorted(agg): print(f"{k:32s} {agg[k]/cnt[k]:16.4g}  (n={cnt[k]})")
