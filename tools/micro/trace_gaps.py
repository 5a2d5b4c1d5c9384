import csv,sys,collections
rows=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0].replace("void ","").replace("ndcn::","")[-34:]) for r in csv.DictReader(open(sys.argv[1])))
lo=int(len(rows)*0.5)
sel=rows[lo:]
busy=sum(e-s for s,e,_ in sel); span=sel[-1][1]-sel[0][0]
print("kernels",len(sel),"span ms",span/1e6,"busy ms",busy/1e6,"idle frac",1-busy/span)
agg=collections.defaultdict(lambda:[0,0.0])
for i in range(len(sel)-1):
    g=(sel[i+1][0]-sel[i][1])/1e3
    if g>15:
        k=sel[i][2]+" -> "+sel[i+1][2]
        agg[k][0]+=1; agg[k][1]+=g
tot=sum(v[1] for v in agg.values())
print("gaps >15us total ms", tot/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:22]:
    print("  %-72s n=%4d total %.2f ms avg %.0f us" % (k, v[0], v[1]/1e3, v[1]/v[0]))
