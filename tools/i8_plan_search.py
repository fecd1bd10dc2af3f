"""Deal (tile, slot range) items of the int8 Gram plan to 8 waves.
NG=6, symmetric diagonal tiles: off-diagonal tile slots = groups 0..5 (1,2,3,4,5,6 MFMAs), diagonal tile slots = Q1..Q5 (1,1,2,2,3), R0,R2,R4 (1,1,1).
Constraints: <= 9 accumulators per wave; SIMD pairs (w, w+4) balanced; few segments (fragment reads)."""
import itertools, random, sys
NG = int(sys.argv[1]) if len(sys.argv) > 1 else 6
SYM = int(sys.argv[2]) if len(sys.argv) > 2 else 1
random.seed(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
def off_slots():
    return [min(k, 5) - max(0, k - 5) + 1 for k in range(NG)]
def diag_slots():
    if not SYM: return off_slots()
    q = []
    for k in range(1, NG):
        q.append(sum(1 for s in range(6) for t in range(6) if s < t and s + t == k))
    r = [1 for k in range(0, NG, 2)]
    return q + r
tiles = [(I, K) for I in range(4) for K in range(I + 1)]
slots = {t: (diag_slots() if t[0] == t[1] else off_slots()) for t in tiles}
total = sum(sum(v) for v in slots.values()); nacc = sum(len(v) for v in slots.values())
print("NG", NG, "sym", SYM, "MFMAs", total, "accs", nacc, "diag slots", diag_slots(), "off", off_slots())

def frags(tile, a, b):
    """distinct (rowblock, slice) fragments needed by slots a..b of tile"""
    I, K = tile
    need = set()
    if I != K or not SYM:
        for k in range(a, b + 1):
            for s in range(6):
                t = k - s
                if 0 <= t <= 5: need.add((I, s)); need.add((K, t))
    else:
        nq = NG - 1
        for q in range(a, b + 1):
            if q < nq:
                k = q + 1
                for s in range(6):
                    t = k - s
                    if s < t <= 5: need.add((I, s)); need.add((I, t))
            else:
                s = q - nq; need.add((I, s))
    return need

def score(assign):
    # assign: list of (wave, tile, a, b)
    load = [0] * 8; acc = [0] * 8; fr = [set() for _ in range(8)]
    for w, t, a, b in assign:
        load[w] += sum(slots[t][a:b + 1]); acc[w] += b - a + 1; fr[w] |= frags(t, a, b)
    pair = [load[w] + load[w + 4] for w in range(4)]
    nfr = sum(len(f) for f in fr)
    imb = max(pair) - min(pair)
    inner = max(abs(load[w] - load[w + 4]) for w in range(4))
    over = sum(max(0, x - 9) for x in acc)
    return (over, max(pair), nfr + 2 * inner, imb), load, acc, pair, nfr

best = None
def random_assign():
    assign = []
    for t in tiles:
        n = len(slots[t])
        # cut into 1-3 segments
        r = random.random()
        nseg = 1 if r < 0.45 else (2 if r < 0.9 else 3)
        cuts = sorted(random.sample(range(1, n), nseg - 1)) if nseg > 1 else []
        bounds = [0] + cuts + [n]
        ws = random.sample(range(8), nseg)
        for i in range(nseg):
            assign.append((ws[i], t, bounds[i], bounds[i + 1] - 1))
    return assign
import time
t0 = time.time()
budget = float(sys.argv[4]) if len(sys.argv) > 4 else 60
it = 0
while time.time() - t0 < budget:
    it += 1
    a = random_assign()
    sc = score(a)
    if sc is None: continue
    # local search: move a segment to another wave
    improved = True
    while improved:
        improved = False
        for i in range(len(a)):
            for w in range(8):
                if w == a[i][0]: continue
                if any(x[1] == a[i][1] and x[0] == w for x in a): continue
                b2 = list(a); b2[i] = (w,) + a[i][1:]
                s2 = score(b2)
                if s2 is not None and s2[0] < sc[0]:
                    a, sc, improved = b2, s2, True
        # shift a cut between two adjacent segments of the same tile
        for i in range(len(a)):
            for j in range(len(a)):
                if i == j or a[i][1] != a[j][1] or a[i][3] + 1 != a[j][2]: continue
                for d in (-1, 1):
                    ni = (a[i][0], a[i][1], a[i][2], a[i][3] + d); nj = (a[j][0], a[j][1], a[j][2] + d, a[j][3])
                    if ni[3] < ni[2] or nj[3] < nj[2]: continue
                    b2 = list(a); b2[i] = ni; b2[j] = nj
                    s2 = score(b2)
                    if s2[0] < sc[0]:
                        a, sc, improved = b2, s2, True
    if best is None or sc[0] < best[1][0]:
        best = (a, sc)
        print(it, sc[0], "load", sc[1], "acc", sc[2], "pair", sc[3], "frags", sc[4]); sys.stdout.flush()
if len(sys.argv) > 5:
    cur = [(0,(0,0),0,6),(0,(2,0),3,4),(1,(1,1),0,6),(1,(3,1),3,4),(2,(3,0),2,6),(2,(1,0),0,3),(3,(3,2),2,6),(3,(2,0),0,2),(3,(2,1),6,6),
           (4,(2,2),0,6),(4,(3,0),0,1),(5,(3,3),0,6),(5,(3,2),0,1),(6,(2,1),0,5),(6,(2,0),5,6),(7,(1,0),4,6),(7,(3,1),5,6),(7,(3,1),0,2)]
    print("current plan:", score(cur))
a, sc = best
for w in range(8):
    print("wave", w, [(t, x, y) for (ww, t, x, y) in sorted(a) if ww == w])
