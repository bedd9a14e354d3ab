import csv, collections, sys
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    agg[(r['Kernel_Name'][:30], r.get('Grid_Size_X'))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:5]:
    print(k, "n",len(v), "avg us %.1f"%(sum(v)/len(v)), "max %.1f"%max(v))
