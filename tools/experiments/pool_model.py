#!/usr/bin/env python3
"""CPU model of the three queues of a persistent-wave path tracer (walk / brick round / transition) with the per-ray statistics measured on the
2048^3 path trace (profiles/r03_cfg4_phase_profile_dilated.txt: 103 trips, 3.18 bricks entered, 0.7 hits per ray; a trip 32 instructions, a brick
round 400 + 30 per voxel trip of its longest lane, a round of transitions 1500): wave-instructions per ray for a ray per lane (vrt_path_kernel),
two rays per lane without exchange between lanes, and a pool of N rays per wave (vrt_pool_kernel).  Written before the kernel (round 4)."""
import random, sys
random.seed(1)
A=0.914; H=0.22; TRIPS=29.6; VOX=6.4
TRIP=32; CALL=60; BRICK_FIX=400; VTRIP=30; TRANS=1500; SWAP=45
def newseg(): 
    # trips until event (geometric)
    import math
    p=1.0/TRIPS
    return max(1,int(math.log(1-random.random())/math.log(1-p))+1)
def voxtrips():
    import math
    p=1.0/VOX
    return min(22,max(1,int(math.log(1-random.random())/math.log(1-p))+1))
W,P,D=0,1,2  # walking, parked, done(waiting transition)
class Ray:
    __slots__=('st','left')
    def __init__(s): s.st=D; s.left=0
def start(r): r.st=W; r.left=newseg()
def single(nrays_total=200000, batch=32, fin=32):
    lanes=[Ray() for _ in range(64)]
    t=0; done_rays=0; stats=dict(calls=0,inl=0,outl=0,br=0,brl=0,tr=0,trl=0,tw=0,tb=0,tt=0)
    while done_rays<nrays_total:
        nW=sum(r.st==W for r in lanes); nD=sum(r.st==D for r in lanes)
        if nD and (nW==0 or nD>=min(fin,max(1,nW//2))):
            t+=TRANS; stats['tt']+=TRANS; stats['tr']+=1; stats['trl']+=nD
            for r in lanes:
                if r.st==D: start(r); done_rays+=1
        nW=sum(r.st==W for r in lanes)
        if nW==0: continue
        f=min(fin,max(1,nW//2)); min_alive=nW-f+1 if nW>=f else 1
        stats['calls']+=1; stats['inl']+=nW
        t+=CALL; stats['tw']+=CALL
        parked=0; trips=0
        while True:
            trips+=1; t+=TRIP; stats['tw']+=TRIP
            alive=0
            for r in lanes:
                if r.st==W:
                    r.left-=1
                    if r.left==0:
                        if random.random()<A: r.st=P; parked+=1
                        else: r.st=D
                    else: alive+=1
            if parked>=batch or alive==0: break
            if trips%4==1 and alive<min_alive: break
        stats['outl']+=alive
        if parked:
            mv=max(voxtrips() for r in lanes if r.st==P)
            c=BRICK_FIX+VTRIP*mv; t+=c; stats['tb']+=c; stats['br']+=1; stats['brl']+=parked
            for r in lanes:
                if r.st==P:
                    if random.random()<H: r.st=D
                    else: r.st=W; r.left=newseg()
    return t/done_rays, stats
def dual(nrays_total=200000, K=20, Bthr=48, Tthr=40, Wmin=24, NC=2):
    # lanes[l] = list of NC rays; index 0 active
    lanes=[[Ray() for _ in range(NC)] for _ in range(64)]
    t=0; done_rays=0; stats=dict(calls=0,inl=0,outl=0,br=0,brl=0,tr=0,trl=0,tw=0,tb=0,tt=0,ts=0)
    def bring(l,st):
        # make active a ray with state st if any; returns True if active has st
        if l[0].st==st: return True
        for i in range(1,NC):
            if l[i].st==st:
                l[0],l[i]=l[i],l[0]; return True
        return False
    while done_rays<nrays_total:
        pW=sum(any(r.st==W for r in l) for l in lanes)
        pB=sum(any(r.st==P for r in l) for l in lanes)
        pT=sum(any(r.st==D for r in l) for l in lanes)
        if pB>=Bthr or (pB and pW<Wmin and pB>=pT):
            t+=SWAP; stats['ts']+=SWAP
            part=[l for l in lanes if bring(l,P)]
            mv=max(voxtrips() for l in part)
            c=BRICK_FIX+VTRIP*mv; t+=c; stats['tb']+=c; stats['br']+=1; stats['brl']+=len(part)
            for l in part:
                r=l[0]
                if random.random()<H: r.st=D
                else: r.st=W; r.left=newseg()
            continue
        if pT>=Tthr or (pT and pW<Wmin):
            t+=SWAP+TRANS; stats['ts']+=SWAP; stats['tt']+=TRANS
            part=[l for l in lanes if bring(l,D)]
            stats['tr']+=1; stats['trl']+=len(part)
            for l in part: start(l[0]); done_rays+=1
            continue
        # walk
        t+=SWAP+CALL; stats['ts']+=SWAP; stats['tw']+=CALL
        part=[l for l in lanes if bring(l,W)]
        nW=len(part); stats['calls']+=1; stats['inl']+=nW
        gone=0; trips=0
        while True:
            trips+=1; t+=TRIP; stats['tw']+=TRIP
            alive=0
            for l in part:
                r=l[0]
                if r.st==W:
                    r.left-=1
                    if r.left==0:
                        gone+=1
                        if random.random()<A: r.st=P
                        else: r.st=D
                    else: alive+=1
            if alive==0: break
            if gone>=K and trips%4==1: break
        stats['outl']+=alive
    return t/done_rays, stats
def show(name,res):
    c,s=res
    print(f"{name}: {c:.1f} wave-instr/ray; walk {100*s['tw']/(s['tw']+s['tb']+s['tt']+s.get('ts',0)):.0f}% brick {100*s['tb']/(s['tw']+s['tb']+s['tt']+s.get('ts',0)):.0f}% trans {100*s['tt']/(s['tw']+s['tb']+s['tt']+s.get('ts',0)):.0f}% swap {100*s.get('ts',0)/(s['tw']+s['tb']+s['tt']+s.get('ts',0)):.0f}% | calls in {s['inl']/s['calls']:.1f} out {s['outl']/s['calls']:.1f} | brick lanes {s['brl']/max(1,s['br']):.1f} | trans lanes {s['trl']/max(1,s['tr']):.1f}")
show("single", single())
for K in (12,16,24):
  for B in (40,56):
    for T in (32,48):
        show(f"dual K{K} B{B} T{T}", dual(K=K,Bthr=B,Tthr=T))
show("triple K16 B56 T48", dual(K=16,Bthr=56,Tthr=48,NC=3))

def pool(nrays_total=200000, N=128, K=16, Bthr=56, Tthr=48, Wmin=32, SW=45):
    rays=[Ray() for _ in range(N)]
    t=0; done_rays=0; stats=dict(calls=0,inl=0,outl=0,br=0,brl=0,tr=0,trl=0,tw=0,tb=0,tt=0,ts=0)
    while done_rays<nrays_total:
        w=[r for r in rays if r.st==W]; b=[r for r in rays if r.st==P]; d=[r for r in rays if r.st==D]
        if len(b)>=Bthr or (b and len(w)<Wmin and len(b)>=len(d)):
            part=b[:64]
            t+=SW; stats['ts']+=SW
            mv=max(voxtrips() for r in part)
            c=BRICK_FIX+VTRIP*mv; t+=c; stats['tb']+=c; stats['br']+=1; stats['brl']+=len(part)
            for r in part:
                if random.random()<H: r.st=D
                else: r.st=W; r.left=newseg()
            continue
        if len(d)>=Tthr or (d and len(w)<Wmin):
            part=d[:64]
            t+=SW+TRANS; stats['ts']+=SW; stats['tt']+=TRANS; stats['tr']+=1; stats['trl']+=len(part)
            for r in part: start(r); done_rays+=1
            continue
        part=w[:64]
        t+=SW+CALL; stats['ts']+=SW; stats['tw']+=CALL
        nW=len(part); stats['calls']+=1; stats['inl']+=nW
        gone=0; trips=0
        while True:
            trips+=1; t+=TRIP; stats['tw']+=TRIP
            alive=0
            for r in part:
                if r.st==W:
                    r.left-=1
                    if r.left==0:
                        gone+=1
                        if random.random()<A: r.st=P
                        else: r.st=D
                    else: alive+=1
            if alive==0: break
            if gone>=K and trips%4==1: break
        stats['outl']+=alive
    return t/done_rays, stats
for N in (128,192,256):
  for K in (8,12,16):
    show(f"pool N{N} K{K}", pool(N=N,K=K, Bthr=60, Tthr=60))
print("---- sensitivity")
for N in (96,112,128,160):
    show(f"pool N{N} K16 SW80", pool(N=N,K=16,Bthr=min(60,N-64+8),Tthr=min(60,N-64+8),SW=80))
for B,T in ((40,40),(48,48),(56,56),(62,62)):
    show(f"pool N128 K16 SW80 B{B} T{T}", pool(N=128,K=16,Bthr=B,Tthr=T,SW=80))
