import csv,glob,re,sys
def short(n):
    n=n.replace("(anonymous namespace)::","").replace("void ","")
    return re.split(r"[<(]",n,maxsplit=1)[0].strip()
path=glob.glob(sys.argv[1]+"/**/*_kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(path)),key=lambda r:int(r["Start_Timestamp"]))
bw=[i for i,r in enumerate(rows) if short(r["Kernel_Name"])=="seg_bwd_persist"]
for i in bw[2:10]:
    q=rows[i]["Queue_Id"]; end=int(rows[i]["End_Timestamp"])
    nxt=[r for r in rows[i+1:i+200] if r["Queue_Id"]==q and short(r["Kernel_Name"]).startswith("gemm")]
    ad=[r for r in rows[i+1:i+300] if short(r["Kernel_Name"])=="adam_kernel"]
    print("bwd %.0f us; end -> next gemm on its queue %.1f us; bwd end -> adam start %.1f us"%((end-int(rows[i]["Start_Timestamp"]))/1e3,(int(nxt[0]["Start_Timestamp"])-end)/1e3,(int(ad[0]["Start_Timestamp"])-end)/1e3))
