"""Monte-Carlo model of the stream upkeep rule of the board sweep (csrc/mcq_hip.hip: LOW_WATER, room_limit): 16 chains of a
wavefront consume words with the step's distribution (two masked-rejection draws, candidates until != old_k, two uniform words);
an upkeep event runs when some chain is below the low-water mark and serves every chain that can take a block.  Prints events
per step and the share of chains served per event, for 16-word blocks (shipped) and for 32-word blocks in the same 64-slot ring.
usage: python tools/sim_upkeep.py > profiles/rNN_upkeep_simulation.txt"""
import numpy as np, sys
rng=np.random.default_rng(1)
def words_per_step(n, N=12, mask=16):
    # i, j: geometric with p=N/mask; k: repeat accepted draws until != old (prob (N-1)/N); + 2
    p=N/mask
    w=rng.geometric(p,size=n)+rng.geometric(p,size=n)
    # k draws
    k=np.zeros(n,int); need=np.ones(n,bool)
    while need.any():
        k[need]+=rng.geometric(p,size=need.sum())
        again=rng.random(n)<1.0/N
        need=need&again
    return w+k+2
def sim(LOW=24, ROOM=53, BLOCK=16, RING=64, C=16, steps=200000, policy="any"):
    avail=np.full(C,40); pending=np.zeros(C,bool)
    events=0; served=0; dry=0; issues=0
    for s in range(steps):
        need=words_per_step(C)
        trig=(avail<LOW).any()
        if trig:
            events+=1
            comp=pending.copy()
            avail[comp]+=BLOCK; pending[comp]=False
            iss=avail<=ROOM
            pending[iss]=True
            served+=max(comp.sum(), iss.sum()); issues+=iss.sum()
        short=need>avail
        if short.any():
            dry+=short.sum()
            # sequential path: service itself: complete pending / issue+complete
            for c in np.where(short)[0]:
                while need[c]>avail[c]:
                    if pending[c]: avail[c]+=BLOCK; pending[c]=False
                    else: avail[c]+=BLOCK
        avail-=need
        assert (avail<=RING).all(), avail.max()
    return events/steps, issues/max(1,events)/C, dry/steps/C
for LOW in (16,20,24,28,32):
    e,part,dry=sim(LOW=LOW,steps=30000)
    print("LOW",LOW,"events/step %.3f participation %.2f dry/step/chain %.4f"%(e,part,dry))

def sim2(LOW=24, ISSUE=40, BLOCK=32, RING=64, C=16, steps=30000):
    avail=np.full(C,40); pending=np.zeros(C,bool)
    events=0; landed=0; issued=0; dry=0
    for s in range(steps):
        need=words_per_step(C)
        if (avail<LOW).any():
            events+=1
            can=pending&(avail<=RING-BLOCK)
            avail[can]+=BLOCK; pending[can]=False; landed+=can.sum()
            iss=(~pending)&(avail<=ISSUE)
            pending[iss]=True; issued+=iss.sum()
        short=need>avail
        if short.any():
            dry+=short.sum()
            for c in np.where(short)[0]:
                while need[c]>avail[c]:
                    avail[c]+=BLOCK; pending[c]=False
        avail-=need
        assert (avail<=RING).all()
    return events/steps, landed/max(1,events)/C, dry/steps/C
print("32-word blocks")
for LOW in (20,24,28):
    for ISSUE in (32,40,48,56,64):
        e,part,dry=sim2(LOW=LOW,ISSUE=ISSUE)
        print("LOW",LOW,"ISSUE",ISSUE,"events/step %.3f landed-participation %.2f dry %.4f"%(e,part,dry))
