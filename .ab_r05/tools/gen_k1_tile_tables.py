"""Tile-to-wave tables of bin_gram_kernel<19> (frank_amd/csrc/bin_gram.hip: kTiles19P0 / kTiles19P1).

The 190 upper-triangle tiles (row-major numbering) are dealt to the 12 waves of a workgroup as compact rectangles (two
block rows x <= 5 block columns) so that a wave reads few distinct 16-column fragments per k-step; <= 10 tiles per wave,
equal tile counts per SIMD (waves W, W+4, W+8 share SIMD W % 4).  Prints the two C tables."""
NBT = 19


def rft(I):
    return I * NBT - I * (I - 1) // 2


def tid(I, J):
    return rft(I) + (J - I)


def rect(rows, c0, c1):
    return [(I, J) for I in rows for J in range(max(c0, I), c1 + 1)]


g = {0: rect([0, 1], 0, 4), 1: rect([0, 1], 5, 9), 2: rect([0, 1], 10, 14), 3: rect([0, 1], 15, 18),
     4: rect([2, 3], 2, 6), 5: rect([2, 3], 7, 11), 6: rect([2, 3], 12, 16),
     7: rect([4, 5], 4, 8), 8: rect([4, 5], 9, 13), 9: rect([4, 5], 14, 18),
     10: rect([6], 6, 13), 11: rect([2, 3], 17, 18) + rect([6], 14, 18)}
assert sorted(sum(g.values(), [])) == sorted((I, J) for I in range(7) for J in range(I, 19))
wave0 = {0: 1, 4: 2, 8: 3, 1: 5, 5: 6, 9: 10, 2: 8, 6: 0, 10: 4, 3: 9, 7: 7, 11: 11}
h = {0: rect([7, 8], 7, 10), 1: rect([7, 8], 11, 13), 2: rect([7, 8], 14, 16), 3: rect([7, 8], 17, 18),
     4: rect([9, 10], 9, 12), 5: rect([9, 10], 13, 15), 6: rect([9, 10], 16, 18),
     7: rect([11, 12], 11, 14), 8: rect([11, 12], 15, 18),
     9: rect([13, 14], 13, 15), 10: rect([13, 14], 16, 18), 11: rect([15, 16, 17, 18], 15, 18)}
assert sorted(sum(h.values(), [])) == sorted((I, J) for I in range(7, 19) for J in range(I, 19))
wave1 = {0: 11, 4: 9, 8: 3, 1: 8, 5: 1, 9: 2, 2: 0, 6: 4, 10: 5, 3: 7, 7: 6, 11: 10}
for name, grp, wm in (("kTiles19P0", g, wave0), ("kTiles19P1", h, wave1)):
    print("constexpr short %s[12][10] = {" % name)
    for W in range(12):
        t = grp[wm[W]]
        ids = [tid(I, J) for (I, J) in t] + [-1] * (10 - len(t))
        print("    {%s},  // wave %d: %d tiles, %d fragments" % (", ".join(map(str, ids)), W, len(t),
                                                                 len(set([a for a, b in t] + [b for a, b in t]))))
    print("};")
    print("// tiles per SIMD:", [sum(len(grp[wm[W]]) for W in (s, s + 4, s + 8)) for s in range(4)])
