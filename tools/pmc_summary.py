import csv, collections, sys
pat = sys.argv[1]
for f in sys.argv[2:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat in k:
            acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k)
        for c, v in d.items():
            print("   %-28s n=%d avg=%.4g" % (c, len(v), sum(v)/len(v)))
