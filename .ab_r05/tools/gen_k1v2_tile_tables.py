#!/usr/bin/env python3
"""Tile-to-wave tables of bin_gram v2 (frank_amd/csrc/bin_gram2.hip).

The NBT x NBT upper triangle of 16x16 Gram tiles is cut into PARTS of at most WAVES * CAP tiles (one workgroup
specialisation each; a part's workgroups generate only the column blocks its tiles touch) and a part's tiles are dealt to
its WAVES waves, at most CAP per wave (CAP * 8 accumulator registers), so that a wave reads few distinct 16-column
fragments per k-step (the union of the row and column blocks of its tiles).  Simulated annealing from a banded start;
deterministic (seeded).  Prints C tables: kTilesNBT<n>[parts][waves][cap] of row-major triangle indices, -1 padded."""
import random
import sys

WAVES, CAP = 8, 24


def tiles_of(NBT):
    return [(I, J) for I in range(NBT) for J in range(I, NBT)]


def tid(NBT, I, J):
    return I * NBT - I * (I - 1) // 2 + (J - I)


def nfrag(ts):
    s = set()
    for I, J in ts:
        s.add(I)
        s.add(J)
    return len(s)


def cost(assign):
    return sum(nfrag(ts) for ts in assign)


def anneal(tiles, nw, cap, seed=1, iters=400000):
    rnd = random.Random(seed)
    # banded start: row-major order cut into equal runs
    per = -(-len(tiles) // nw)
    assign = [list(tiles[w * per:(w + 1) * per]) for w in range(nw)]
    best, bestc = [list(a) for a in assign], cost(assign)
    cur = bestc
    T = 2.0
    for it in range(iters):
        T = max(0.02, 2.0 * (1 - it / iters))
        a, b = rnd.randrange(nw), rnd.randrange(nw)
        if a == b or not assign[a]:
            continue
        ia = rnd.randrange(len(assign[a]))
        if len(assign[b]) < cap and rnd.random() < 0.5:  # move
            t = assign[a][ia]
            before = nfrag(assign[a]) + nfrag(assign[b])
            assign[a].pop(ia)
            assign[b].append(t)
            after = nfrag(assign[a]) + nfrag(assign[b])
            if after <= before or rnd.random() < pow(2.718, -(after - before) / T):
                cur += after - before
            else:
                assign[b].pop()
                assign[a].insert(ia, t)
        elif assign[b]:  # swap
            ib = rnd.randrange(len(assign[b]))
            before = nfrag(assign[a]) + nfrag(assign[b])
            assign[a][ia], assign[b][ib] = assign[b][ib], assign[a][ia]
            after = nfrag(assign[a]) + nfrag(assign[b])
            if after <= before or rnd.random() < pow(2.718, -(after - before) / T):
                cur += after - before
            else:
                assign[a][ia], assign[b][ib] = assign[b][ib], assign[a][ia]
        if cur < bestc:
            bestc, best = cur, [list(x) for x in assign]
    return best


def split_parts(NBT, part_cap):
    """Row-aligned parts of <= part_cap tiles, balanced in tile count (a part streams every visibility)."""
    tiles = tiles_of(NBT)
    nparts = -(-len(tiles) // part_cap)
    target = len(tiles) / nparts
    parts, cur = [], []
    for I in range(NBT):
        row = [(I, J) for J in range(I, NBT)]
        if cur and len(cur) + len(row) > part_cap or (cur and len(parts) < nparts - 1 and len(cur) + len(row) / 2 > target):
            parts.append(cur)
            cur = []
        cur = cur + row
    parts.append(cur)
    assert all(len(p) <= part_cap for p in parts), [len(p) for p in parts]
    return parts


def main():
    for NBT in [int(x) for x in sys.argv[1:]] or [4, 8, 13, 19, 24, 32]:
        parts = split_parts(NBT, WAVES * CAP)
        print("// NBT = %d: %d tiles, %d part(s) of %s tiles" % (NBT, len(tiles_of(NBT)), len(parts), [len(p) for p in parts]))
        print("constexpr short kTiles%d[%d][%d][%d] = {" % (NBT, len(parts), WAVES, CAP))
        for P, pt in enumerate(parts):
            a = anneal(pt, WAVES, min(CAP, -(-len(pt) // WAVES)), seed=NBT * 10 + P)  # balanced: every wave ceil(n/8)
            # waves w and w + 4 share a SIMD: order the waves so that the SIMD totals are as even as possible
            a.sort(key=len, reverse=True)
            order = [a[0], a[2], a[4], a[6], a[7], a[5], a[3], a[1]]
            print("  {  // part %d: block rows %d..%d, column blocks %d..%d" % (
                P, min(t[0] for t in pt), max(t[0] for t in pt), min(t[0] for t in pt), NBT - 1))
            for w, ts in enumerate(order):
                ts = sorted(ts)
                ids = [tid(NBT, I, J) for I, J in ts] + [-1] * (CAP - len(ts))
                print("    {%s},  // wave %d: %d tiles, %d fragments" % (", ".join(map(str, ids)), w, len(ts), nfrag(ts)))
            print("  },")
            print("  // tiles per SIMD: %s" % [len(order[s]) + len(order[s + 4]) for s in range(4)])
        print("};")


if __name__ == "__main__":
    main()
