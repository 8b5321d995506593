import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
seq=[]
for r in rows:
    n=r["Kernel_Name"]
    if "gemm_bf16_kernel" in n:
        seq.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
out=[]
for i in range(0,len(seq),45):
    d=sorted(seq[i:i+45]); out.append("%.1f" % d[len(d)//2])
print(sys.argv[2], " ".join(out))
