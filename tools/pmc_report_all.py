"""Sum of every counter per kernel name (first dispatch of each kernel) from rocprofv3 --pmc csv output."""
import collections
import csv
import glob
import sys

per = collections.OrderedDict()
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    first = {}
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if sys.argv[2] not in name:
            continue
        d = int(r["Dispatch_Id"])
        first.setdefault(name, d)
        if d != first[name]:
            continue
        k = per.setdefault(name, collections.OrderedDict())
        k["_meta"] = "grid %s vgpr %s sgpr %s lds %s" % (r["Grid_Size"], r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"])
        k[r["Counter_Name"]] = k.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for name, k in per.items():
    print(name, k.pop("_meta"))
    for c, v in k.items():
        print("   %-28s %18.0f" % (c, v))
