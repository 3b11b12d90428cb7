"""CPU mirrors of two integer schedules the kernels compute from device-side counts (no GPU needed): they pin the arithmetic the
comments in csrc/k_mlp.hip (k_part_mlp_all: one part per workgroup) and csrc/k_knn.hip (k_pair_lists: list offsets from the
per-group pair counts) describe, over random and degenerate count vectors."""
import numpy as np

P = 5
PER_BLOCK = 128          # (MLP_BLOCK / 64) * MLP_CB * 16
PAIR_GROUP = 4096


def mlp_ranges(counts, n_linear, G):
    """csrc/k_mlp.hip:k_part_mlp_all — [(part, first workgroup, number of workgroups, stride)] per workgroup b."""
    tiles = [((c + PER_BLOCK - 1) // PER_BLOCK) * (8 if nl == 3 else 5) for c, nl in zip(counts, n_linear)]
    total = sum(tiles)
    served = {b: [] for b in range(G)}
    if total == 0:
        return served
    cum = 0
    for p in range(P):
        start, end = cum * G // total, (cum + tiles[p]) * G // total
        cum += tiles[p]
        if tiles[p] == 0:
            continue
        if end > start:
            for b in range(start, end):
                served[b].append((p, b - start, end - start))
        else:
            served[min(start, G - 1)].append((p, 0, 1))
    return served


def test_every_mlp_tile_is_served_exactly_once():
    rng = np.random.default_rng(0)
    cases = [[1162270, 1379147, 615803, 721528, 769595], [145000, 172000, 77000, 90000, 96000], [51324, 1, 11574, 1, 10810],
             [0, 0, 0, 0, 0], [1, 1, 1, 1, 1], [0, 5, 0, 0, 70000], [129, 128, 127, 1, 0]]
    cases += [list(rng.integers(0, 10 ** rng.integers(1, 7), P)) for _ in range(200)]
    for counts in cases:
        for G in (1, 3, 16, 768):
            nl = [3, 2, 2, 3, 2]
            served = mlp_ranges([int(c) for c in counts], nl, G)
            seen = [np.zeros((int(c) + PER_BLOCK - 1) // PER_BLOCK, np.int32) for c in counts]
            for b, lst in served.items():
                assert len({p for p, _, _ in lst}) == len(lst)
                for p, vb, nb in lst:
                    t = vb
                    while t * PER_BLOCK < counts[p]:          # mlp_part: tiles vb, vb + nb, ...
                        seen[p][t] += 1
                        t += nb
            for p in range(P):
                assert (seen[p] == 1).all(), (counts, G, p)
            if G == 768 and min(counts) > 100 * PER_BLOCK:      # large parts: one weight staging per workgroup
                assert max(len(v) for v in served.values()) == 1


def test_pair_list_offsets_give_ascending_dense_lists():
    """csrc/k_knn.hip:k_pair_lists — a group's list offset is the sum of the counts of the groups before it; inside a group the
    ranks follow the slot order."""
    rng = np.random.default_rng(1)
    for na in (0, 1, 63, 4096, 4097, 300000):
        flags = rng.integers(0, 32, na).astype(np.uint8) * (rng.random(na) < 0.6)
        flags = flags.astype(np.uint8)
        n_groups = (max(na, 1) - 1) // PAIR_GROUP + 1
        gcount = np.zeros((n_groups, P), np.int64)
        for p in range(P):                                        # what k_knn_pairs accumulates per ticket of 64
            bit = (flags >> p) & 1
            for g in range(n_groups):
                gcount[g, p] = bit[g * PAIR_GROUP:(g + 1) * PAIR_GROUP].sum()
        lists = [np.full(int(gcount[:, p].sum()) + 1, -7, np.int64) for p in range(P)]
        for g in range(n_groups):
            base = gcount[:g].sum(0)
            for p in range(P):
                sl = np.nonzero((flags[g * PAIR_GROUP:(g + 1) * PAIR_GROUP] >> p) & 1)[0] + g * PAIR_GROUP
                lists[p][base[p]:base[p] + sl.size] = sl
        for p in range(P):
            want = np.nonzero((flags >> p) & 1)[0]
            assert np.array_equal(lists[p][:-1], want)            # dense, ascending, every flagged slot once


# ---- round 3: the colour kernel's segment cursor (csrc/k_mlp.hip: seg_window / seg_locate / rgb_part) ----------------------------
def rgb_tiles_by_cursor(wcnt, gcount, n_waves):
    """Mirror of k_part_rgb_all for ONE part: the winners of slot group g sit at list positions [pairs before g, + wcnt[g]); a
    wave's unit is a tile of 32 winners of one segment; tile T -> (segment, offset) through a cursor over windows of 64
    segments with inclusive tile / exclusive pair prefix sums.  -> [(list position, valid) x 32] per tile, in tile order, as the
    waves (T = wave, wave + n_waves, ...) produce them."""
    g_last = len(wcnt) - 1
    total = sum((c + 31) >> 5 for c in wcnt)
    out = {}
    for w in range(n_waves):
        g0, tiles_before, pairs_before = 0, 0, 0

        def window(g0):
            wc = [wcnt[g] if g <= g_last else 0 for g in range(g0, g0 + 64)]
            gc = [gcount[g] if g <= g_last else 0 for g in range(g0, g0 + 64)]
            tl = [(c + 31) >> 5 for c in wc]
            ti = list(np.cumsum(tl))
            pe = list(np.cumsum(gc) - np.array(gc))
            return wc, tl, ti, pe, int(ti[-1]), int(sum(gc))
        wc, tl, ti, pe, win_tiles, win_pairs = window(g0)
        for T in range(w, total, n_waves):
            while T >= tiles_before + win_tiles and g0 + 64 <= g_last:
                tiles_before += win_tiles
                pairs_before += win_pairs
                g0 += 64
                wc, tl, ti, pe, win_tiles, win_pairs = window(g0)
            L = next(l for l in range(64) if tiles_before + ti[l] > T)
            t0, sb, sn = tiles_before + ti[L] - tl[L], pairs_before + pe[L], wc[L]
            cols = []
            for j in range((T - t0) * 32, (T - t0) * 32 + 32):
                cols.append((sb + min(j, sn - 1), j < sn))
            out[T] = cols
    return [out[T] for T in range(total)]


def test_colour_kernel_segment_cursor_visits_every_winner_once():
    rng = np.random.default_rng(3)
    for n_groups, n_waves in ((1, 4), (3, 7), (64, 12), (65, 5), (130, 3072), (584, 3072), (700, 33)):
        gcount = rng.integers(0, 4097, n_groups)
        wcnt = np.array([int(rng.integers(0, g + 1)) if rng.random() > 0.15 else 0 for g in gcount])
        wcnt[-1] += 1                                            # the far-constant pair rides in the last segment
        off = np.cumsum(gcount) - gcount
        want = [int(off[g]) + r for g in range(n_groups) for r in range(int(wcnt[g]))]
        tiles = rgb_tiles_by_cursor([int(x) for x in wcnt], [int(x) for x in gcount], n_waves)
        got = [pos for cols in tiles for pos, valid in cols if valid]
        assert got == want, (n_groups, n_waves)
        # invalid columns re-read a winner of the SAME segment (a valid address, its result is not stored)
        for cols in tiles:
            seg_lo = min(pos for pos, _ in cols)
            assert all(seg_lo <= pos <= seg_lo + 31 for pos, _ in cols)
        assert len(tiles) == sum((int(c) + 31) >> 5 for c in wcnt)


# ---- round 3: the merge rule of k_winner_lists against the oracle's merge (inb_part_network_multiassign.py:229-256) --------------
def wsel_mirror(occ, listed, far, occ_const):
    """csrc/k_mlp.hip:k_winner_lists per survivor: candidates = listed occupancy / the part's far constant / 0 for an unflagged part;
    start from part 0 whatever it is, a later part takes over only with a strictly larger occupancy.
    -> sel: p (listed pair of part p), 8 + p (far constant of part p), 255 (zeros)."""
    n = occ.shape[0]
    sel = np.full(n, 255, np.int64)
    best = np.zeros(n, np.float32)
    for p in range(5):
        c = np.where(listed[:, p], occ[:, p], np.where(far[:, p], occ_const[p], np.float32(0.0))).astype(np.float32)
        s = np.where(listed[:, p], p, np.where(far[:, p], 8 + p, 255))
        take = np.ones(n, bool) if p == 0 else c > best
        best = np.where(take, c, best)
        sel = np.where(take, s, sel)
    return sel


def test_winner_rule_equals_the_oracles_merge():
    import torch
    from oracle import nvr_oracle as O          # checker only
    rng = np.random.default_rng(11)
    n = 20000
    occ = rng.random((n, 5)).astype(np.float32)
    occ[rng.random((n, 5)) < 0.15] = 0.0                                     # exact zeros and
    occ[:, 3] = np.where(rng.random(n) < 0.3, occ[:, 1], occ[:, 3])          # exact ties between parts
    rgb = rng.random((n, 5, 3)).astype(np.float32)
    kind = rng.integers(0, 3, (n, 5))                                         # 0 unflagged, 1 listed, 2 far
    listed, far = kind == 1, kind == 2
    occ_const = rng.random(5).astype(np.float32)
    rgb_const = rng.random((5, 3)).astype(np.float32)
    # the dense (N, 5, 4) tensor the reference merges: zeros for unflagged parts (:199-200,229-233), the part constant for far pairs
    raws = np.zeros((n, 5, 4), np.float32)
    raws[..., :3] = np.where(listed[..., None], rgb, np.where(far[..., None], rgb_const[None], 0.0))
    raws[..., 3] = np.where(listed, occ, np.where(far, occ_const[None], 0.0))
    want_raw, want_occ = O.merge_parts(torch.from_numpy(raws))
    sel = wsel_mirror(occ, listed, far, occ_const)
    got = np.zeros((n, 4), np.float32)
    for p in range(5):
        m = sel == p
        got[m, :3], got[m, 3] = rgb[m, p], occ[m, p]
        m = sel == 8 + p
        got[m, :3], got[m, 3] = rgb_const[p], occ_const[p]
    assert np.array_equal(got, want_raw.numpy())
    assert np.array_equal(got[:, 3], want_occ.numpy().reshape(-1))
    assert (sel == 255).sum() > 100 and (sel < 5).sum() > 1000 and ((sel >= 8) & (sel < 16)).sum() > 1000


# ---- round 3: the x-corner rows of a hashed level by a +-delta fold (csrc/k_encode.hip: level_rowsum, GridDev.xdelta) -------------
def test_hashed_x_corner_delta_fold_equals_the_direct_hash():
    """hash(cx, cy, cz) = (cx * 1) ^ (cy * 19349663) ^ (cz * 83492791) mod T (part_base_embedder.py:132-136).  x's prime is 1, so the
    key of the c1x corner is the c0x corner's key X with the bits m = c0x ^ c1x flipped: X ^ m = X + m - 2 (X & m); its residue is
    the c0x corner's + that delta, folded once into [0, T) — for every cell a level can have (res <= 8192) and every table size in
    use (T = nextprime(2^k) > 2^14)."""
    P1, P2 = 19349663, 83492791
    rng = np.random.default_rng(21)
    for T in (32771, 262147, 1048583):
        for res in (23, 250, 1291, 2004, 8192):
            c0x = rng.integers(0, res, 4000)
            d = rng.choice([0, 1, 1, 1, 2], 4000)                      # clipped / regular / the f + 1 rounding case (:116)
            c1x = np.minimum(c0x + d, res - 1)
            cy, cz = rng.integers(0, res, 4000), rng.integers(0, res, 4000)
            X = c0x.astype(object) ^ (cy.astype(object) * P1) ^ (cz.astype(object) * P2)
            r0 = np.array([int(v) % T for v in X], dtype=np.int64)
            want = np.array([int((int(a) ^ (int(b) * P1) ^ (int(c) * P2)) % T) for a, b, c in zip(c1x, cy, cz)], dtype=np.int64)
            m = (c0x ^ c1x).astype(np.int64)
            xlo = np.array([int(v) & 0xFFFFFFFF for v in X], dtype=np.int64)
            r1 = r0 + m - 2 * (xlo & m)
            r1 = np.where(r1 < 0, r1 + T, r1)
            r1 = np.where(r1 >= T, r1 - T, r1)
            assert np.array_equal(r1, want), (T, res)


def test_two_round_hash_reduction_equals_the_modulo():
    """common.h:hash_mod24_2r — (key mod T) for T = 2^k + c by two folding rounds 2^k == -c (mod T) and one fix-up each way, valid when
    c * h1 < 2^k (invr_abi.hip:make_grid_dev checks it on worst-case bounds; asserted here on the values): integer mirror against
    Python's % on random 40-bit keys and on the extremes, for every table size the x-corner fold is enabled for (k >= 18)."""
    rng = np.random.default_rng(33)
    for T in (262147, 524309, 1048583, 2097169, 4194319):
        k = T.bit_length() - 1
        c = T - (1 << k)
        mask = (1 << k) - 1
        keys = np.concatenate([rng.integers(0, 1 << 40, 200000, dtype=np.int64), np.array([0, 1, T - 1, T, T + 1, (1 << 40) - 1, (1 << 41) - 1], dtype=np.int64),
                               (np.arange(1, 2000, dtype=np.int64) * T) - 1, np.arange(1, 2000, dtype=np.int64) * T])
        a0, h0 = keys & mask, keys >> k
        assert int(h0.max()) < 1 << 24 and c < 1 << 24                      # operands of v_mul_u32_u24
        y1 = c * h0
        a1, h1 = y1 & mask, y1 >> k
        assert int((c * h1).max()) < 1 << k                                 # the second fold lands below 2^k: no third round
        v = a0 + c * h1 - a1
        assert int(v.min()) > -(1 << k) and int(v.max()) < 2 * T
        v = np.where(v < 0, v + T, v)
        v = np.where(v >= T, v - T, v)
        assert np.array_equal(v, keys % T), T
