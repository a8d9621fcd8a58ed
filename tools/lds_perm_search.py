import sys, itertools
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from lds_model import radices, ways

def bitperm(t, perm):
    # perm: tuple of source bit for each destination bit (6 bits within the wave); upper bits kept
    lo = t & 63; hi = t & ~63; out = 0
    for d, s in enumerate(perm):
        out |= ((lo >> s) & 1) << d
    return hi | out

def pass_cost(N, E, p, f, perm):
    T = N // E; rad = radices(N, E)
    Ns = 1
    for q in range(p): Ns *= rad[q]
    R = rad[p]; NB = E // R
    tw = tr = 0
    nw = max(1, T // 64)
    for w in range(nw):
        lanes = [64 * w + l for l in range(min(64, T))]
        js = [bitperm(t, perm) for t in lanes]
        if p < len(rad) - 1:
            for b in range(NB):
                for r in range(R):
                    arr = [f(((j + T*b) // Ns) * (Ns * R) + ((j + T*b) % Ns) + r * Ns) for j in js]
                    tw += sum(ways(arr[g:g + 16], 32) for g in range(0, 64, 16))
        if p > 0:
            for e in range(E):
                arr = [f(j + T * e) for j in js]
                tr += sum(ways(arr[g:g + 32], 64) for g in range(0, 64, 32))
    return tw / nw, tr / nw

pads = {"pad3": lambda i: i + (i >> 3), "pad4": lambda i: i + (i >> 4), "pad5": lambda i: i + (i >> 5), "pad6": lambda i: i + (i >> 6), "none": lambda i: i,
        "pad5x2": lambda i: i + 2 * (i >> 5), "pad4+8": lambda i: i + (i >> 4) + (i >> 8), "pad5+8": lambda i: i + (i >> 5) + (i >> 8)}
perms = list(itertools.permutations(range(6)))
for N, E in ((2048, 8), (4096, 16)):
    rad = radices(N, E)
    print("N", N, "E", E, rad)
    for pname, f in pads.items():
        total = 0; desc = []
        for p in range(len(rad)):
            best = None
            for perm in perms:
                w, r = pass_cost(N, E, p, f, perm)
                c = w + r
                if best is None or c < best[0]:
                    best = (c, w, r, perm)
                    ideal = (4 * E if p < len(rad) - 1 else 0) + (2 * E if p > 0 else 0)
                    if c == ideal: break
            total += best[0]; desc.append(best)
        ideal_total = sum((4 * E if p < len(rad) - 1 else 0) + (2 * E if p > 0 else 0) for p in range(len(rad)))
        print(f"  {pname:8s} total {total:6.1f} (ideal {ideal_total})", [(d[0], d[3]) for d in desc])
