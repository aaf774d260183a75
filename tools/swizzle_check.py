"""Brute-force check of the LDS read addresses of the 32x32x16 layout tried
on the persistent conv in round 3 (profiles/r03/README.md; the layout is
restated below — the variant itself is not kept in the tree): for every
ds_read_b128 the
16 lanes of each of the instruction's four lane groups (MI355X_MICROARCH.md
§LDS) must touch 16 distinct 16-byte bank slots (byte address / 16 mod 16),
for every tap shift, k-step, fragment and consumer wave."""
H0, H1, H2, TS1, MFW = 6, 10, 18, 8, 4
SLAB_OFF = H0 * H1 * H2 * 128
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
          [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS = GROUPS + [[l + 32 for l in g] for g in GROUPS]


def slots(addrs):
    return [(a // 16) % 16 for a in addrs]


def check(name, addr_of_lane):
    for g in GROUPS:
        s = slots([addr_of_lane(l) for l in g])
        assert len(set(s)) == 16, (name, sorted(s))


n = 0
for wave in range(8):
    mf0 = wave * MFW
    row0 = (mf0 // TS1) * H1 + (mf0 % TS1)
    for tc in range(3):
        for tb in range(3):
            for ta in range(3):
                for nt in range(2):
                    for kp in range(4):
                        def pos(l):
                            j, h = l & 31, l >> 5
                            jt, js = j & 15, j >> 4
                            x = tb & 1
                            base = ((row0 + js) * H2 + jt + tc) * 128 + \
                                ((h ^ ((jt + tc) & 7) ^ ((js ^ x) & 1)) << 4)
                            return (base ^ (kp << 5)) + ta * H1 * H2 * 128 + \
                                ((2 * nt + tb) * H2) * 128
                        check(('pos', wave, tc, tb, ta, nt, kp), pos)
                        n += 1
                        # the address equals the producer's layout: cell * 128
                        # + ((chunk ^ key) << 4), key = (c2 & 7) ^ (c1 & 1)
                        for l in range(64):
                            j, h = l & 31, l >> 5
                            c1 = (mf0 % TS1) + (j >> 4) + 2 * nt + tb
                            c2 = (j & 15) + tc
                            c0 = mf0 // TS1 + ta
                            cell = (c0 * H1 + c1) * H2 + c2
                            want = cell * 128 + (((2 * kp + h) ^ (c2 & 7) ^ (c1 & 1)) << 4)
                            assert pos(l) == want, (l, pos(l), want)
for mt in range(2):
    for kp in range(4):
        for slot in range(3):
            def filt(l):
                j, h = l & 31, l >> 5
                rho = mt * 32 + j
                base = SLAB_OFF + rho * 128 + ((h ^ ((rho >> 1) & 7)) << 4)
                return (base ^ (kp << 5)) + slot * 8192
            check(('filter', mt, kp, slot), filt)
            n += 1
print('conflict-free:', n, 'read patterns')
