import csv,glob,re,sys
def short(n):
    n=n.replace("(anonymous namespace)::","").replace("void ","")
    return re.split(r"[<(]",n,maxsplit=1)[0].strip()
path=glob.glob(sys.argv[1]+"/**/*_kernel_trace.csv",recursive=True)[0]
which=int(sys.argv[2]) if len(sys.argv)>2 else 5
thr=float(sys.argv[3]) if len(sys.argv)>3 else 30.0
rows=sorted(csv.DictReader(open(path)),key=lambda r:int(r["Start_Timestamp"]))
adam=[i for i,r in enumerate(rows) if short(r["Kernel_Name"])=="adam_kernel"]
a,b=adam[which],adam[which+1]
t0=int(rows[a]["End_Timestamp"])
print("step %.3f ms"%((int(rows[b]["End_Timestamp"])-t0)/1e6))
for r in rows[a+1:b+1]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    d=(e-s)/1e3
    nm=short(r["Kernel_Name"])
    if (s-t0)/1e3>float(sys.argv[4] if len(sys.argv)>4 else 2000) and (d>=thr or nm.startswith(("seg","adam","seam"))):
        print("%8.1f %8.1f  q%s %-32s" % ((s-t0)/1e3,d,r["Queue_Id"],nm[:32]))
