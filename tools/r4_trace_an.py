import sys
import numpy as np
for fn in sys.argv[1:]:
    v=[int(x) for x in open(fn)]
    # split by markers (0)
    segs=[];cur=[]
    for x in v:
        if x==0:
            if cur: segs.append(cur)
            cur=[]
        else: cur.append(x)
    if cur: segs.append(cur)
    print(fn,"stretches",len(segs),"lens",[len(s) for s in segs][:12])
    d=np.concatenate([np.diff(s) for s in segs if len(s)>1])
    print("  iteration cycles: n=%d mean=%.0f median=%.0f p10=%.0f p90=%.0f p99=%.0f max=%.0f"%(len(d),d.mean(),np.median(d),np.percentile(d,10),np.percentile(d,90),np.percentile(d,99),d.max()))
    big=d[d>2*np.median(d)]
    print("  iterations > 2x median: %d (%.1f%%), their share of time %.1f%%"%(len(big),100*len(big)/len(d),100*big.sum()/d.sum()))
    gaps=[segs[i+1][0]-segs[i][-1] for i in range(len(segs)-1)]
    print("  gap between stretches (last iteration top -> first of next):",[int(g) for g in gaps][:10])
    # print a sample window of deltas
    print("  sample:", [int(x) for x in np.diff(segs[0])[:40]])
