import json, sys
from collections import defaultdict
rows=[json.loads(l) for l in open(sys.argv[1])]
agg=defaultdict(lambda:[0,0.0,0.0])
for r in rows:
    if r['kernel']=='gemm_conv':
        key=(r['M'],r['N'],r['K'],r['taps'],r['stride'],r['up'],r['geglu'],r['cat'])
    else:
        key=('attn',r['B'],r['heads'],r['Nq'],r['Nkv'])
    a=agg[key]; a[0]+=1; a[1]+=r['us']; a[2]=r['tflops']
tot=sum(a[1] for a in agg.values())
g=sum(a[1] for k,a in agg.items() if k[0]!='attn')
print(f"total {tot:.0f} us  gemm {g:.0f} us  attn {tot-g:.0f} us")
n=int(sys.argv[2]) if len(sys.argv)>2 else 30
for k,a in sorted(agg.items(), key=lambda kv:-kv[1][1])[:n]:
    print(f"{str(k):58s} n={a[0]:3d} tot={a[1]:8.1f}us ({100*a[1]/tot:4.1f}%) avg={a[1]/a[0]:7.1f}us {a[2]:6.0f} TF")
