import sys, random
sys.path[:0]=['.','oracle','tests']
import numpy as np
import tracy_amd, pyoracle as orc
SC=(3,-5,-10,-4)
rng=random.Random(101)
def rand_seq(n): return bytes(rng.choice(b"ACGT") for _ in range(n))
ctx=tracy_amd.Context(0)
def ends_of(btr,n):
    fwd=btr[::-1]; return len(fwd)-len(fwd.lstrip(b'h')), n-(len(fwd)-len(fwd.rstrip(b'h')))
for single in (False, True):
    a1=[];a2=[];lo=[];hi=[];w=[]
    for it in range(300):
        m=rng.randint(1,400); a=rand_seq(m)
        b=rand_seq(rng.randint(0,60))+a+rand_seq(rng.randint(0,60))
        ws,wb=orc.gotoh_str(a,b,1,0,SC); n=len(b)
        lead,ce=ends_of(wb,n); g=rng.randint(0,30); d1=ce-m
        a1.append(a);a2.append(b);lo.append(d1-g-1);hi.append(d1+g+1);w.append((ws,wb))
    if single:
        res=[ctx.align_banded([a1[i]],[a2[i]],SC+(1,0),[lo[i]],[hi[i]]) for i in range(len(a1))]
        sc=[r[0][0] for r in res]; btr=[r[1][0] for r in res]
    else:
        sc,btr=ctx.align_banded(a1,a2,SC+(1,0),lo,hi)
    bad=[(i,len(a1[i]),len(a2[i]),hi[i]-lo[i], int(sc[i])==w[i][0], len(btr[i])) for i in range(len(a1)) if (int(sc[i]),btr[i])!=w[i]]
    print("single" if single else "batch", len(bad), bad[:12])
