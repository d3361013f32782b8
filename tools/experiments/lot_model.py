# Monte Carlo of one persistent wave's scheduler (phase costs from tools/tune.py --census) with a parking lot of CAP paths:
# cost per ray, lanes per walking step, lanes per shade phase.  usage: python lot_model.py <gamma shape of the steps-per-segment distribution>
import random, sys
random.seed(1); SH=2.5
C_STEP, C_SCHED, C_LEAF, C_SHADE, C_SWAP = 50, 52, 110, 1013, 90
def new_seg():
    # steps until finished, with leaf stops at random step positions
    n = max(1, int(random.gammavariate(SH, 33.2/SH)+0.5))
    s = 0
    # number of sphere tests ~ Poisson-ish mean 2.14
    x = random.random(); 
    import math
    # poisson sample
    L = math.exp(-2.14); k=0; p=1.0
    while True:
        p *= random.random()
        if p <= L: break
        k += 1
    stops = sorted(random.randint(1, n) for _ in range(k))
    return [n, stops]   # remaining steps, stops measured in remaining steps from start
class Path:
    __slots__=('left','stops','done','state')
    def __init__(s):
        n, st = new_seg(); s.left=n; s.stops=[n - x for x in st]  # stop when left == value
        s.stops.sort(reverse=True); s.state='run'
def sim(CAP, shade_num, leaf_num=2, run_shift=1, refill_min=1, ncyc=3000):
    lanes=[None]*64   # None => needs shade (vacant)
    lotW=[]; lotS=0
    cost=0; rays=0; walk_steps=0; walk_lanes=0; shade_lanes=0; shades=0
    # states: Path.state in run/leaf ; None => shade wait
    while shades < ncyc:
        # refill
        vac=[i for i in range(64) if lanes[i] is None]
        if lotW and len(vac)>=refill_min:
            t=min(len(vac),len(lotW))
            for i in vac[:t]:
                lanes[i]=lotW.pop(); lotS+=1
            cost+=C_SWAP
        n_run=sum(1 for p in lanes if p is not None and p.state=='run')
        n_leaf=sum(1 for p in lanes if p is not None and p.state=='leaf')
        n_shade=64-n_run-n_leaf
        cost+=C_SCHED
        n_ns=n_run+n_leaf
        shade_tot=n_shade+min(lotS,n_ns)
        if n_leaf*16>=64*leaf_num and n_leaf>0: ph='leaf'
        elif shade_tot*16>=64*shade_num and shade_tot>0: ph='shade'
        elif n_run>0: ph='run'
        else: ph='leaf' if n_leaf>=shade_tot else 'shade'
        if ph=='leaf':
            cost+=C_LEAF
            for p in lanes:
                if p is not None and p.state=='leaf':
                    p.state='run'
        elif ph=='shade':
            # swap non-shading with S entries
            ns=[i for i in range(64) if lanes[i] is not None]
            t=min(len(ns),lotS)
            if t or (len(lotW)+lotS<CAP and ns): cost+=C_SWAP
            for i in ns[:t]:
                lotW.append(lanes[i]); lanes[i]=None; lotS-=1
            # push
            room=CAP-len(lotW)-lotS
            for i in ns[t:t+room]:
                lotW.append(lanes[i]); lanes[i]=None
            cnt=0
            for i in range(64):
                if lanes[i] is None:
                    lanes[i]=Path(); cnt+=1
            cost+=C_SHADE; rays+=cnt; shades+=1; shade_lanes+=cnt
        else:
            leave_at=n_run-max(1,n_run>>run_shift)
            while True:
                cnt=0
                for i,p in enumerate(lanes):
                    if p is not None and p.state=='run':
                        cnt+=1
                        p.left-=1
                        if p.left<=0: lanes[i]=None
                        elif p.stops and p.stops[-1]>=p.left:
                            p.stops.pop(); p.state='leaf'
                cost+=C_STEP; walk_steps+=1; walk_lanes+=cnt
                nr=sum(1 for p in lanes if p is not None and p.state=='run')
                if nr<=leave_at: break
    return cost/rays, walk_lanes/walk_steps, shade_lanes/shades, walk_steps/rays
SH=float(sys.argv[1])
for CAP in (0,6,8,16):
    for sn in (12,13,14,15):
        r=sim(CAP,sn)
        print(CAP,sn,'cost/ray %.1f walk lanes %.1f shade lanes %.1f steps/ray %.3f'%r)
