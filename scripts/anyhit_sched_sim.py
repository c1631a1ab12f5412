#!/usr/bin/env python
"""Scheduling model of the lean any-hit walk (k_shadow_anyhit, fujiyama-renderer_amd/csrc/device/fjgpu_dev_anyhit.h): how many wave-level
instructions a ray costs under three ways of binding rays to lanes, on a synthetic stream of rays with the walk's own statistics
(19.8 inner-node steps per ray, ~2.5 leaves of 1-4 triangles, one ray in eight occluded, lengths correlated along the queue).

  A   the shipped walk: one ray per lane, the wave runs ONE phase per iteration (turnover / inner / leaf), turnover once `refill`
      lanes are idle, up to 8 inner steps per vote, postponed leaves
  A2  VERDICT round 4, design (a): TWO rays bound to every lane, a lane takes part in a phase if either of its rays waits there
      (+19 v_cndmask per inner step to select / write back the state, +6 per leaf step)
  B   design (b): a wave owns a POOL of NS ray slots in LDS, lanes are workers: every iteration the wave picks up to 64 slots that
      wait in the same phase (+16 instructions of selection, +6 of address arithmetic per step; the state loads are DS instructions)

Output per variant: wave-level instructions per ray, loop iterations, and executions x mean active lanes of the inner (I), leaf (L)
and turnover (T) phases.  Costs: inner step 100, triangle test 110, turnover 260, vote 18 (counted from the kernel's ISA).
The model is MORE divergent than the chip (A: 29.5 lanes per inner step against 37.6 measured), so it overstates what filling buys;
what it is used for is the relation between the variants and the number of rays a wave must hold.  profiles/r05_anyhit_lane_filling.txt
reads the numbers together with the LDS budget.  usage: python scripts/anyhit_sched_sim.py [rays]"""
import random, math, sys
random.seed(1)
# ray model: list of ops: 'I' inner step, 'L' one triangle test (leaf of k tris = k consecutive L's unless hit)
CL = dict(left=0, mean=19.8)
SIG_C, SIG_R, CLN = 0.7, 0.25, 64
def make_ray():
    if CL['left'] == 0:
        CL['left'] = CLN
        CL['mean'] = random.lognormvariate(math.log(19.8) - 0.5*SIG_C**2 - 0.5*SIG_R**2, SIG_C)
    CL['left'] -= 1
    n = max(1, int(random.lognormvariate(math.log(CL['mean']), SIG_R) + .5))
    ops = []
    # leaves: on average 2.5 leaves per ray, each 1..4 tris (avg 2.2) ; place at random positions after some inner step
    nleaf = min(n, max(0, int(random.gauss(2.5 * n / 19.8, 1.0) + .5)))
    pos = sorted(random.randrange(n) for _ in range(nleaf))
    leaves = {}
    for p in pos:
        leaves.setdefault(p, []).append(random.choice([1, 2, 2, 3, 3, 4, 1, 2]))
    for i in range(n):
        ops.append('I')
        for k in leaves.get(i, []):
            ops.append(('L', k))
    occluded = random.random() < 0.125
    if occluded:
        cut = random.randrange(1, len(ops) + 1)
        ops = ops[:cut]
    return ops

COST = dict(I=100, L=110, T=260, vote=18)

class RayState:
    __slots__ = ('ops', 'i', 'pleaf', 'ltris')
    def __init__(s, ops):
        s.ops = ops; s.i = 0; s.pleaf = 0; s.ltris = 0
    # state: fin if i>=len and pleaf==0 and ltris==0

def simA(nrays=200000, refill=40, max_steps=8, min_inner=16, lanes=64):
    """current scheme. per-lane state: cur op; postponed leaf"""
    total_rays = 0
    rays = [None] * lanes
    instr = 0; iters = 0
    ph = dict(I=[0, 0], L=[0, 0], T=[0, 0])
    remaining = nrays
    def fin(r): return r is None or (r.i >= len(r.ops) and r.pleaf == 0 and r.ltris == 0)
    def at_inner(r): return r is not None and r.i < len(r.ops) and r.ops[r.i] == 'I' and r.ltris == 0
    def at_leaf(r): return r is not None and (r.pleaf > 0 or r.ltris > 0)
    def advance_to_state(r):
        # after an inner step: if next op is a leaf: postpone if possible (stack not empty ~ more ops follow) else become leaf
        while r.i < len(r.ops) and r.ops[r.i] != 'I' and r.ltris == 0:
            k = r.ops[r.i][1]
            more_after = r.i + 1 < len(r.ops)
            if r.pleaf == 0 and more_after:
                r.pleaf = k; r.i += 1
            else:
                r.ltris = k; r.i += 1
    while True:
        iters += 1
        instr += COST['vote']
        nf = sum(1 for r in rays if fin(r) and (r is not None or remaining > 0))
        ni = sum(1 for r in rays if at_inner(r))
        nl = sum(1 for r in rays if at_leaf(r))
        if nf >= refill or (ni == 0 and nl == 0):
            if nf == 0: break
            cnt = 0
            for k in range(lanes):
                if fin(rays[k]):
                    if rays[k] is not None: rays[k] = None; cnt += 1
                    if remaining > 0:
                        remaining -= 1; rays[k] = RayState(make_ray()); cnt += (0 if cnt else 0)
            ph['T'][0] += 1; ph['T'][1] += nf
            instr += COST['T']
            if remaining == 0 and all(r is None for r in rays): break
            continue
        if ni >= nl:
            for step in range(max_steps):
                act = [r for r in rays if at_inner(r)]
                if step > 0 and len(act) < min_inner: break
                if step > 0: instr += 6
                ph['I'][0] += 1; ph['I'][1] += len(act); instr += COST['I']
                for r in act:
                    r.i += 1
                    advance_to_state(r)
        else:
            act = [r for r in rays if at_leaf(r)]
            ph['L'][0] += 1; ph['L'][1] += len(act); instr += COST['L']
            for r in act:
                if r.pleaf > 0: r.pleaf -= 1
                else:
                    r.ltris -= 1
                    if r.ltris == 0: advance_to_state(r)
    return instr, iters, ph

def simB(nrays=200000, NS=80, t_turn=20, t_leaf=32, lanes=64, max_steps=1, min_inner=48, sel=16, ldsov=6):
    """pool: NS slots per wave; each iteration run the phase with most ready slots (capped 64)"""
    slots = [None] * NS
    remaining = nrays
    instr = 0; iters = 0
    ph = dict(I=[0, 0], L=[0, 0], T=[0, 0])
    def fin(r): return r is None or (r.i >= len(r.ops) and r.ltris == 0)
    def at_inner(r): return r is not None and r.ltris == 0 and r.i < len(r.ops) and r.ops[r.i] == 'I'
    def at_leaf(r): return r is not None and r.ltris > 0
    def settle(r):
        if r.ltris == 0 and r.i < len(r.ops) and r.ops[r.i] != 'I':
            r.ltris = r.ops[r.i][1]; r.i += 1
    while True:
        iters += 1
        instr += COST['vote'] + sel
        F = [k for k in range(NS) if fin(slots[k]) and (slots[k] is not None or remaining > 0)]
        I = [k for k in range(NS) if at_inner(slots[k])]
        L = [k for k in range(NS) if at_leaf(slots[k])]
        if not F and not I and not L: break
        # choose
        ni, nl, nf = min(len(I), lanes), min(len(L), lanes), min(len(F), lanes)
        if nf >= t_turn or (ni == 0 and nl == 0):
            todo = F[:lanes]
            ph['T'][0] += 1; ph['T'][1] += len(todo); instr += COST['T'] + ldsov
            for k in todo:
                slots[k] = None
                if remaining > 0:
                    remaining -= 1; slots[k] = RayState(make_ray()); settle(slots[k])
            continue
        if nl >= t_leaf or ni == 0 or nl >= ni:
            todo = L[:lanes]
            ph['L'][0] += 1; ph['L'][1] += len(todo); instr += COST['L'] + ldsov
            for k in todo:
                r = slots[k]; r.ltris -= 1
                if r.ltris == 0: settle(r)
            continue
        todo = I[:lanes]
        for step in range(max_steps):
            act = [k for k in todo if at_inner(slots[k])]
            if step > 0 and len(act) < min_inner: break
            ph['I'][0] += 1; ph['I'][1] += len(act); instr += COST['I'] + (ldsov if step == 0 else 4)
            for k in act:
                r = slots[k]; r.i += 1; settle(r)
    return instr, iters, ph


def simA2(nrays=200000, refill=40, max_steps=8, min_inner=16, lanes=64, sel_inner=19, sel_leaf=6):
    """design (a): TWO rays bound to every lane (state in registers, selected per lane by v_cndmask); a lane takes part in a phase if either
    of its rays waits there"""
    rays = [[None, None] for _ in range(lanes)]
    remaining = nrays
    instr = 0; iters = 0
    ph = dict(I=[0, 0], L=[0, 0], T=[0, 0])
    def fin(r): return r is None or (r.i >= len(r.ops) and r.pleaf == 0 and r.ltris == 0)
    def at_inner(r): return r is not None and r.i < len(r.ops) and r.ops[r.i] == 'I' and r.ltris == 0
    def at_leaf(r): return r is not None and (r.pleaf > 0 or r.ltris > 0)
    def advance(r):
        while r.i < len(r.ops) and r.ops[r.i] != 'I' and r.ltris == 0:
            k = r.ops[r.i][1]
            if r.pleaf == 0 and r.i + 1 < len(r.ops): r.pleaf = k; r.i += 1
            else: r.ltris = k; r.i += 1
    def pick(pair, pred):
        for r in pair:
            if pred(r): return r
        return None
    while True:
        iters += 1
        instr += COST['vote'] + 8
        nf_slots = sum(1 for p in rays for r in p if fin(r) and (r is not None or remaining > 0))
        nf_lanes = sum(1 for p in rays if any(fin(r) and (r is not None or remaining > 0) for r in p))
        ni = sum(1 for p in rays if pick(p, at_inner))
        nl = sum(1 for p in rays if pick(p, at_leaf))
        if nf_slots >= refill or (ni == 0 and nl == 0):
            if nf_slots == 0: break
            for p in rays:
                for k in (0, 1):
                    if fin(p[k]) and (p[k] is not None or remaining > 0):
                        p[k] = None
                        if remaining > 0:
                            remaining -= 1; p[k] = RayState(make_ray())
                        break                      # one slot per lane and execution
            ph['T'][0] += 1; ph['T'][1] += nf_lanes; instr += COST['T'] + 10
            if remaining == 0 and all(r is None for p in rays for r in p): break
            continue
        if ni >= nl:
            for step in range(max_steps):
                act = [pick(p, at_inner) for p in rays]
                act = [r for r in act if r is not None]
                if step > 0 and len(act) < min_inner: break
                ph['I'][0] += 1; ph['I'][1] += len(act); instr += COST['I'] + sel_inner
                for r in act:
                    r.i += 1; advance(r)
        else:
            act = [pick(p, at_leaf) for p in rays]
            act = [r for r in act if r is not None]
            ph['L'][0] += 1; ph['L'][1] += len(act); instr += COST['L'] + sel_leaf
            for r in act:
                if r.pleaf > 0: r.pleaf -= 1
                else:
                    r.ltris -= 1
                    if r.ltris == 0: advance(r)
    return instr, iters, ph

def show(name, res, nrays):
    instr, iters, ph = res
    s = "%-34s instr/ray %.1f iters %d" % (name, instr / nrays * 64 / 64, iters)
    for k in 'ILT':
        n, l = ph[k]
        s += "  %s %d x %.1f" % (k, n, l / max(1, n))
    print(s, flush=True)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
random.seed(1); show("A refill40 steps8 min16", simA(N), N)
random.seed(1); show("A refill24", simA(N, refill=24), N)
for rf in (32, 48, 64):
    random.seed(1); show("A2 two rays per lane, refill %d slots" % rf, simA2(N, refill=rf), N)
for NS, tt, tl in ((80, 20, 32), (80, 16, 40), (96, 24, 40), (128, 40, 56), (128, 32, 64), (160, 48, 64), (64, 24, 24)):
    random.seed(1); show("B NS=%d turn%d leaf%d" % (NS, tt, tl), simB(N, NS=NS, t_turn=tt, t_leaf=tl), N)
print()
for NS, tt, tl in ((72, 16, 32), (80, 20, 32), (96, 24, 40), (128, 40, 56)):
    random.seed(1); show("B NS=%d turn%d leaf%d steps4/min40" % (NS, tt, tl), simB(N, NS=NS, t_turn=tt, t_leaf=tl, max_steps=4, min_inner=40), N)
