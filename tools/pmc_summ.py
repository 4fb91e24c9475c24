import csv,collections,sys,glob
for d in sys.argv[1:]:
    for f in glob.glob(d+"/*/*_counter_collection.csv"):
        agg=collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "at::" in k or "Cijk" in k: continue
            agg[k[:60]][r["Counter_Name"]]+=float(r["Counter_Value"])
        for k,dd in agg.items():
            print(d,k)
            wc=dd.get("SQ_WAVE_CYCLES",0) or 1
            for c,val in sorted(dd.items()):
                print("    %-30s %16.0f  %.3f"%(c,val,val/wc))
