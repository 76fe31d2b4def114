import csv, sys, collections, glob
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    acc = collections.OrderedDict()
    for r in rows:
        if sys.argv[2] in r["Kernel_Name"]:
            k = (r["Dispatch_Id"], r["Counter_Name"])
            acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
            meta = (r["Grid_Size"], r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"])
    last = max(int(k[0]) for k in acc) if acc else 0
    print(f, "grid/vgpr/sgpr/lds", meta if acc else None)
    for k, v in acc.items():
        if int(k[0]) == last:
            print("   %-28s %16.0f" % (k[1], v))
