import csv, collections, sys, glob
for f in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "oct_fused_kernel" in k and "false>" in k.split("oct_fused_kernel")[1][:40]:
            acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k)
        for c, v in d.items():
            print("   %-28s n=%d avg=%.4g" % (c, len(v), sum(v)/len(v)))
