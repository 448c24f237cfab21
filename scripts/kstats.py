import csv,glob,sys
for f in glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "filter_pixel_kernel" in r["Name"]: print(r["Name"][35:95], r["Calls"], round(float(r["AverageNs"])/1e6,2))
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    acc={}
    for r in csv.DictReader(open(f)):
        if "filter_pixel_kernel" in r["Kernel_Name"]:
            k=(r["Kernel_Name"][35:80], r["Counter_Name"]); acc.setdefault(k,[]).append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(k, "%.3g GB" % (sum(v)/len(v)*1024/1e9))
