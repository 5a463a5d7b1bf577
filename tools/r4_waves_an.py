import sys, numpy as np
for fn in sys.argv[1:]:
    a=np.loadtxt(fn)
    w,t0,t1,loop,it=a.T
    T0=t0.min(); dur=t1.max()-T0
    print(fn,"waves",len(w),"kernel ticks %.3g"%dur)
    print("  start spread (ticks): p50 %.3g max %.3g"%(np.median(t0-T0),(t0-T0).max()))
    end=(t1-T0)/dur
    print("  wave end / kernel end: p1 %.3f p10 %.3f p50 %.3f p90 %.3f max %.3f"%tuple(np.percentile(end,[1,10,50,90,100])))
    print("  iterations per wave: min %d p10 %d p50 %d p90 %d max %d  sum %.4g"%(it.min(),*np.percentile(it,[10,50,90]),it.max(),it.sum()))
    cpi=loop/np.maximum(it,1)
    print("  loop ticks per iteration per wave: p10 %.0f p50 %.0f p90 %.0f max %.0f"%tuple(np.percentile(cpi,[10,50,90,100])))
    print("  share of wave lifetime in the loop: p50 %.3f"%np.median(loop/(t1-t0)))
    # by block residue: blocks are dealt to XCDs round-robin; per CU grouping unknown -- show early vs late blocks
    for lo,hi in ((0,1024),(1024,2048),(2048,3072),(3072,4096),(4096,5120),(5120,6144)):
        m=(w>=lo)&(w<hi)
        if m.any(): print("   waves %4d-%4d: iterations mean %.0f  ticks/iter %.0f  of which waiting for the text %.0f"%(lo,hi,it[m].mean(),cpi[m].mean(),(t0[m]/np.maximum(it[m],1)).mean()))
