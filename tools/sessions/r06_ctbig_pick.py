#!/usr/bin/env python3
"""How the schedule table of csrc/ctbig_sizes.h (8193 .. 16384 points) was made -- session notes in runnable form, no GPU needed for steps 1-2.

1. candidates(): per 7-smooth size the factor triples (512 threads, at most 50 points per thread and pass) and quadruples (640 .. 1024 threads, at most
   20 .. 30 points) out of the radices fft_lds.h has butterflies for, ranked by a count of LDS instructions per transform; every ordering.
2. probe(): each candidate compiled alone (hipcc -Rpass-analysis=kernel-resource-usage): spilled registers and LDS bytes; candidates with more than
   32 .. 40 spills or more than 160 KiB dropped.
3. Up to twelve candidates per size built into variant libraries (-DMDSP_CTBIG_LIST_H=<list>) and timed on the GPU with tools/sweep_ctcols.py
   (sessions r06s26 .. r06s29); the fastest per size is the table.  profiles/r06_ctbig_lean.json holds every measurement.
"""
import itertools, math, re, subprocess, json, sys
RADS=[2,3,4,5,6,7,8,9,10,12,14,15,16,18,20,21,24,25,27,28,30,32]
def smooth(lo,hi):
    out=[]
    for n in range(lo,hi+1):
        m=n
        for p in (2,3,5,7):
            while m%p==0: m//=p
        if m==1: out.append(n)
    return out
def util(N,T,rs):
    u=0
    for r in rs:
        nbf=N//r; M=-(-nbf//T); u+=nbf/(M*T)
    return u/len(rs)
def pts(N,T,rs): return max((-(-(N//r)//T))*r for r in rs)
def cands(N):
    res=[]
    # three passes, 512 threads, one butterfly per thread and pass
    for rs in itertools.product(RADS,repeat=3):
        if rs[0]*rs[1]*rs[2]!=N: continue
        if any(N//r>512 for r in rs): continue
        res.append((3,512,rs,util(N,512,rs)))
    for T in (640,768,896,1024):
        for rs in itertools.product([r for r in RADS if r<=16],repeat=4):
            if math.prod(rs)!=N: continue
            if pts(N,T,rs)>16: continue
            res.append((4,T,rs,util(N,T,rs)))
    return res

import itertools, math, re, json, sys, subprocess, os
from concurrent.futures import ThreadPoolExecutor
def M_(N,T,r): return -(-(N//r)//T)
def pts(N,T,rs): return max(M_(N,T,r)*r for r in rs)
def ldscost(N,T,rs):
    w=T//64; c=0
    for p,r in enumerate(rs):
        M=M_(N,T,r); c+=w*(M*(2*r+(2*(r-1) if p>0 else 0))+40)
    return c
def NP(N,rs):
    e=0; ns=1
    for p,r in enumerate(rs):
        G=ns*r
        if p<len(rs)-1 and G%4==0: e=max(e,N//G)
        ns*=r
    return N+e
def lds_ok(N,rs): return NP(N,rs)*8+ (128+ -(-N//128))*8 <= 160*1024
def multisets(N):
    out=[]
    for rs in itertools.combinations_with_replacement(RADS,3):
        if math.prod(rs)==N and pts(N,512,rs)<=50: out.append((ldscost(N,512,rs),512,rs))
    for T,cap in ((640,30),(768,30),(896,20),(1024,20)):
        for rs in itertools.combinations_with_replacement([r for r in RADS if r<=16],4):
            if math.prod(rs)==N and pts(N,T,rs)<=cap: out.append((ldscost(N,T,rs),T,rs))
    return sorted(out)
def orders(rs):
    rs=sorted(rs)
    if len(rs)==3:
        a,b,c=rs   # a<=b<=c
        return [(b,a,c),(c,a,b),(c,b,a),(a,b,c)]
    a,b,c,d=rs
    return [(d,c,b,a),(a,b,c,d),(c,a,b,d),(d,a,b,c),(b,c,d,a)]
def probe(N,T,rs):
    tag=f"{N}_{T}_"+"_".join(map(str,rs))
    cmd=["/opt/rocm/bin/hipcc","--offload-arch=gfx950","-O3","-std=c++17","-fno-gpu-rdc","-Wno-unused-function","-ffp-contract=on","-I/opt/rocm/include","-Idsp.jl_amd/csrc",
         f"-DTRY_SIZES(X)=X({N},{T},FL,{','.join(map(str,rs))})","--cuda-device-only","-c","tools/sessions/r06_ctbig_probe.hip","-o",f"/tmp/{tag}.o","-Rpass-analysis=kernel-resource-usage"]
    r=subprocess.run(cmd,capture_output=True,text=True)
    sp=re.search(r"VGPRs Spill: (\d+)",r.stderr); lds=re.search(r"LDS Size \[bytes/block\]: (\d+)",r.stderr); vg=re.search(r" VGPRs: (\d+)",r.stderr)
    os.remove(f"/tmp/{tag}.o") if os.path.exists(f"/tmp/{tag}.o") else None
    if not sp: return (N,T,rs,None,None,None)
    return (N,T,rs,int(sp.group(1)),int(lds.group(1)),int(vg.group(1)))

if __name__ == "__main__":
    for N in smooth(8193, 16384):
        ms = multisets(N)
        print(N, [(T, rs) for _, T, rs in ms[:4]])
