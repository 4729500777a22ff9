import csv,glob,sys,collections
for d in sys.argv[1:]:
    for f in glob.glob(d+"/**/*_counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items():
            if "ewa_" not in k: continue
            print(k)
            for c,vals in sorted(v.items()):
                print("   %-28s n=%d mean=%.4g" % (c,len(vals),sum(vals)/len(vals)))
