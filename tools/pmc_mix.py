import csv, glob, sys, collections
out=sys.argv[1]
files = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")[:80]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
rows=[]
for k,d in acc.items():
    mf=d.get("SQ_INSTS_MFMA",0); va=d.get("SQ_INSTS_VALU",0); sa=d.get("SQ_INSTS_SALU",0); lds=d.get("SQ_INSTS_LDS",0)
    rows.append((va+sa, k, mf, va, sa, lds))
rows.sort(reverse=True)
print("%-82s %10s %10s %10s %10s  %s"%("kernel","MFMA(M)","VALU(M)","SALU(M)","LDS(M)","(VALU-MFMA)/MFMA, SALU/MFMA"))
for t,k,mf,va,sa,lds in rows[:28]:
    print("%-82s %10.2f %10.2f %10.2f %10.2f  %5.2f %5.2f"%(k,mf/1e6,va/1e6,sa/1e6,lds/1e6,(va-mf)/mf if mf else 0, sa/mf if mf else 0))
