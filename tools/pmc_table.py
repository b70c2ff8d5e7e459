"""Average counter values per kernel from a rocprofv3 counter_collection.csv."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    a = acc[k][r["Counter_Name"]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in acc.items():
    print(k)
    for c, (n, v) in sorted(cs.items()):
        print("   %-28s %14.1f  (n=%d)" % (c, v / n, n))
