"""CPU simulation of one wave of k_eval_general settling a whole run of allowed requests per round (max-plus prefix scan as
predictor, the real step as the judge) against the plain sequence: the experiment of profiles/r03_v41_general_allowed_runs.txt."""
import random
I64MAX=(1<<63)-1; I64MIN=-(1<<63); U64=(1<<64)-1
def sat(x): return max(I64MIN,min(I64MAX,x))
def wrap(x): return x & U64
def s64(x):
    x&=U64
    return x-(1<<64) if x>>63 else x
def step(c, ei,dvt,q,now, full=True):
    tat0,exp=c
    live = exp>now
    tat = max(tat0, sat(now-dvt)) if live else sat(now-ei)
    inc=sat(ei*q); new=sat(tat+inc); allow_at=sat(new-dvt)
    allowed = now>=allow_at
    out=c
    if allowed:
        ttl = wrap(sat(sat(new-now)+dvt)); e=now+ttl
        if e>U64: e=U64
        out=(new,e)
    cur = new if allowed else tat
    room=sat(now+dvt-cur)
    rem = max(int(room/ei) if ei>0 else 0,0) if ei>0 else 0
    return allowed, out, rem
def cell_after(new,dvt,now):
    ttl = wrap(sat(sat(new-now)+dvt)); e=now+ttl
    if e>U64: e=U64
    return (new,e)
def wave(c0, reqs):
    n=len(reqs); fin=[False]*n; c=[c0]*n; res=[None]*n; mine=[None]*n; was=[False]*n
    rounds=0
    while not all(fin):
        rounds+=1
        allow=[(not fin[l]) and step(c[l],*reqs[l])[0] for l in range(n)]
        first=next((l for l in range(n) if allow[l]),64)
        pb=[wrap(sat(r[0]*r[2])) for r in reqs]
        A=[wrap(wrap(sat(r[3]-r[1]))+pb[l]) for l,r in enumerate(reqs)]; B=pb[:]
        off=1
        while off<64:
            Ae=[A[l-off] if l-off>=0 else A[l] for l in range(n)]; Be=[B[l-off] if l-off>=0 else B[l] for l in range(n)]
            nA=A[:]; nB=B[:]
            for l in range(n):
                if l>=first+off:
                    x=s64(Ae[l]+B[l]); y=s64(A[l]); nA[l]=wrap(max(x,y)); nB[l]=wrap(Be[l]+B[l])
            A,B=nA,nB; off*=2
        pred=[]
        for l in range(n):
            x=s64(wrap(c[l][0])+B[l]); y=s64(A[l]); pred.append(cell_after(max(x,y),reqs[l][1],reqs[l][3]))
        cin=[c[l] if not (l>first) else pred[l-1] for l in range(n)]
        d=[None]*n; cw=[None]*n; cons=[False]*n
        for l in range(n):
            if not fin[l]:
                a,o,rem=step(cin[l],*reqs[l]); d[l]=(a,rem); cw[l]=o
                cons[l]= a and o==pred[l]
        bad=[l for l in range(n) if l>=first and not cons[l]]
        brk=bad[0] if bad else 64
        if brk<64:
            takes=(not fin[brk]) and d[brk][0]
            nstate = cw[brk] if takes else cin[brk]
        for l in range(n):
            if not fin[l] and (l<first or l<=brk):
                res[l]=d[l]
                if d[l][0]: was[l]=True; mine[l]=cw[l]
                c[l]=cin[l]; fin[l]=True
        if brk<64:
            for l in range(brk+1,n): c[l]=nstate
    out = mine[n-1] if was[n-1] else c[n-1]
    return res,out,rounds
def seq(c0,reqs):
    c=c0; res=[]
    for r in reqs:
        a,o,rem=step(c,*r); res.append((a,rem)); c=o
    return res,c
random.seed(1)
T0=1700000000*10**9
worst=0
for trial in range(20000):
    n=random.choice([1,2,5,17,64])
    ei=random.choice([10**8,6*10**9,1]); dvt=ei*random.choice([0,1,4,99])
    now=T0
    reqs=[]
    for l in range(n):
        now+=random.choice([0,1,10**7,10**9,-10**8,5*10**10])
        if random.random()<0.2:
            e2=random.choice([10**8,3*10**9]); d2=e2*random.choice([0,2,9])
        else: e2,d2=ei,dvt
        reqs.append((e2,d2,random.choice([1,1,1,2,0,3]),max(now,1)))
    c0=random.choice([(0,0),(T0-10**9,T0+10**9),(T0+5*10**9,T0+9*10**9),(T0-10**11,T0-10**10)])
    r1,o1,rounds=wave(c0,reqs); r2,o2=seq(c0,reqs)
    worst=max(worst,rounds)
    if r1!=r2 or o1!=o2:
        print("MISMATCH trial",trial,n,c0); 
        for l in range(n): 
            if r1[l]!=r2[l]: print(l,r1[l],r2[l],reqs[l])
        print(o1,o2); break
else: print("all equal; worst rounds",worst)
