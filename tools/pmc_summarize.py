import csv, glob, sys, collections
d = sys.argv[1]
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "wgrad" not in k and "igemm" not in k: continue
        print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
