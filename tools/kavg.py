import sys, pandas as pd
t=pd.read_csv(sys.argv[1]); t['short']=t.Name.str.extract(r'(\w+_kernel(?:<[^>]*>)?)')[0]
for k in ('potf2_kernel','trsm_blk_kernel','gemm_k64_kernel<1, 1>'):
    r=t[t.short==k]
    if len(r): print(sys.argv[2], k, int(r.Calls.iloc[0]), round(r.AverageNs.iloc[0]/1e3,2))
