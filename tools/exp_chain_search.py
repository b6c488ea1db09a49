#!/usr/bin/env python3
"""Search for a cheap signed-digit exponentiation chain for the BN parameter u on the cyclotomic subgroup (conjugation free, a
squaring = 1458 and a product = 4374 multiply-adds per lane in the lane-pair mapping): for every digit set {1} + up to three odd
digits below 130, a dynamic-programming recoder finds the fewest non-zero digits, a greedy builder prices the table.  Takes ~30 min.
Result used by tools/gen_device_constants.py: digits +-17, +-35 -> 62 squarings + 13 products (width-4 NAF: 63 + 16)."""
import itertools, functools, sys
sys.setrecursionlimit(100000)
u=4965661367192848881
SQ=1458; MU=4374
def recode_cost(k, D):
    mx=max(D)
    @functools.lru_cache(None)
    def f(k):
        if k==0: return (0, ())
        if k%2==0:
            c,ds=f(k//2); return (c, tuple((p+1,d) for p,d in ds))
        best=None
        for d in D:
            for s in (d,-d):
                r=k-s
                if r<0 or r>=2*k: continue
                if r>k and k<4*mx: continue
                c,ds=f(r)
                cand=(c+1, ((0,s),)+ds)
                if best is None or cand[0]<best[0]: best=cand
        return best
    return f(k)
def table_cost(D):
    known={1}; muls=0; sq=0; order=[]
    todo=sorted(set(D)-{1})
    while todo:
        made=False
        for d in list(todo):
            found=None
            for a in sorted(known):
                for b in sorted(known):
                    for j in range(0,8):
                        if d in (a*(1<<j)+b, a*(1<<j)-b, b-a*(1<<j)): found=(a,j,b); break
                    if found: break
                if found: break
            if found:
                known.add(d); todo.remove(d); muls+=1; sq+=found[1]; order.append((d,found)); made=True
        if not made:
            h=min(x for x in range(3,64,2) if x not in known)
            known.add(h); muls+=1; sq+=1; order.append((h,'helper'))
    return muls, sq, order
best=[]
odds=[x for x in range(3,130,2)]
for r in range(0,4):
    for extra in itertools.combinations(odds, r):
        D=(1,)+extra
        res=recode_cost(u,D)
        if res is None: continue
        c,ds=res
        assert sum(d<<p for p,d in ds)==u
        tm,tsq,order=table_cost(D)
        top=max(p for p,d in ds)
        nmul=c-1+tm
        nsq=top+tsq
        cost=nmul*MU+nsq*SQ
        best.append((cost,nmul,nsq,D,c,tm,tsq))
best.sort()
for b in best[:15]: print(b)
print("w4 reference: 16 mul 63 sqr cost", 16*MU+63*SQ)
b=best[0]
print(recode_cost(u,b[3]), table_cost(b[3]))
