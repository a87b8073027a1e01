"""The MSM's split tail (kyber_amd/csrc/msm.cuh: reduce_coop_kernel with `runs` / `fuse`, tree_fold_bits_coop_kernel,
final_rows_kernel's chain table; DESIGN.md section 5 item 58 b) as an integer model: buckets are integers modulo a prime
instead of points, "addition" is +, "doubling" is * 2.  The model follows the kernels' INDEX arithmetic -- which group
adds which at every tree level, which output row a group's slot goes to, how many doublings chain b gets -- so that the
identity the tail rests on,

    sum_b (b + 1) B_b  =  sum_j W_j + chunk * sum_k 2^k D_k,     D_k = sum of the T_j whose chunk number has bit k,

and the row bookkeeping between the launches are checked for every plan shape the planner can produce (chunks of 1..8
buckets, 1..4096 chunks per window, fold groups of 32 or 64, with and without the reduce kernel's own four levels).
No GPU: this is host logic."""
import random

import pytest

P = (1 << 61) - 1
FUSED_BITS = 4  # REDUCE_FUSED_BITS: log2 of reduce_coop_kernel's 16 groups


def reduce_kernel(buckets, nwin, nb, chunk, fuse):
    """reduce_coop_kernel(split): returns (rows, ncur, nplain) in tree_fold_bits_coop_kernel's layout"""
    nchunks = nb // chunk
    W = [[0] * nchunks for _ in range(nwin)]
    T = [[0] * nchunks for _ in range(nwin)]
    for w in range(nwin):
        for ch in range(nchunks):
            lo = ch * chunk
            run = tot = buckets[w][lo + chunk - 1]
            for s in range(chunk - 1):  # steps 2s (run += B), 2s + 1 (tot += run)
                run = (run + buckets[w][lo + chunk - 2 - s]) % P
                tot = (tot + run) % P
            W[w][ch], T[w][ch] = tot, run
    if not fuse:
        return W + T, nchunks, nwin  # rows: W (nwin), then T (nwin) = the bits rows
    G = 16
    ncur = nchunks // G
    rows = [[0] * ncur for _ in range(nwin * (1 + FUSED_BITS) + nwin)]
    for w in range(nwin):
        for c0 in range(ncur):
            run = [T[w][c0 * G + g] for g in range(G)]
            tot = [W[w][c0 * G + g] for g in range(G)]
            for k in range(FUSED_BITS):
                off = 1 << k
                new_run, new_tot = list(run), list(tot)
                for gi in range(G):
                    low = gi & (2 * off - 1)
                    t_act = (low & (low - 1)) == 0 and low < off
                    w_act = low == 2 * off - 1
                    assert not (t_act and w_act)
                    if t_act:
                        new_run[gi] = (run[gi] + run[gi + off]) % P
                    if w_act:
                        new_tot[gi] = (tot[gi] + tot[gi - off]) % P
                run, tot = new_run, new_tot
            rows[w][c0] = tot[G - 1]
            rows[nwin * (1 + FUSED_BITS) + w][c0] = run[0]
            for m in range(FUSED_BITS):
                rows[nwin + w * FUSED_BITS + m][c0] = run[1 << m]
    return rows, ncur, nwin * (1 + FUSED_BITS)


def fold_launch(rows, nplain, nbits, nin, lb_out, FG):
    """tree_fold_bits_coop_kernel: one launch over rows of nin entries"""
    nout = (nin + FG - 1) // FG
    out = [[0] * nout for _ in range(nplain + nbits * (lb_out + 1))]
    for row in range(nplain + nbits):
        bits = row >= nplain
        for g in range(nout):
            v = [rows[row][g * FG + gi] if g * FG + gi < nin else 0 for gi in range(FG)]
            have = nin - g * FG
            off = 1
            while off < FG and off < have:
                nv = list(v)
                for gi in range(FG):
                    low = gi & (2 * off - 1)
                    active = ((low & (low - 1)) == 0 and low < off) if bits else low == 0
                    if active:
                        nv[gi] = (v[gi] + v[gi + off]) % P
                v = nv
                off <<= 1
            if not bits:
                out[row][g] = v[0]
            else:
                w = row - nplain
                out[nplain + nbits * lb_out + w][g] = v[0]
                for m in range(lb_out):
                    if (1 << m) < FG:
                        out[nplain + w * lb_out + m][g] = v[1 << m]
    return out, nout


def chain_doublings(b, nwin, c, lb0, lb, tz, chbits):
    """final_rows_kernel: doublings of chain b"""
    if b < nwin:
        return b * c
    q = b - nwin
    if q < nwin * lb0:
        ww, k = divmod(q, lb0)
    else:
        q -= nwin * lb0
        k0 = lb0
        lbl = min(lb, chbits - k0)
        while q >= nwin * lbl:
            q -= nwin * lbl
            k0 += lbl
            lbl = min(lb, chbits - k0)
        ww, m = divmod(q, lbl)
        k = k0 + m
    return ww * c + tz + k


def split_tail(buckets, nwin, c, nb, chunk, FG, fuse_ok=True):
    nchunks = nb // chunk
    chbits, tz, LB = nchunks.bit_length() - 1, chunk.bit_length() - 1, FG.bit_length() - 1
    fuse = fuse_ok and nchunks >= 16
    lb0 = FUSED_BITS if fuse else 0
    rows, ncur, nplain = reduce_kernel(buckets, nwin, nb, chunk, fuse)
    done = lb0
    while ncur > 1:
        lb_out = min(LB, chbits - done)
        rows, nout = fold_launch(rows, nplain, nwin, ncur, lb_out, FG)
        nplain += nwin * lb_out
        done += lb_out
        ncur = nout
    assert done == chbits and nplain == nwin * (1 + chbits)
    total = 0
    for b in range(nplain):
        total = (total + rows[b][0] * pow(2, chain_doublings(b, nwin, c, lb0, LB, tz, chbits), P)) % P
    return total


def direct(buckets, nwin, c, nb):
    return sum(pow(2, c * w, P) * sum((b + 1) * buckets[w][b] for b in range(nb)) for w in range(nwin)) % P


@pytest.mark.parametrize("FG", [32, 64])
@pytest.mark.parametrize("c,chunk", [(4, 1), (4, 8), (5, 8), (6, 2), (8, 8), (9, 4), (10, 8), (13, 8), (14, 8)])
def test_split_tail_equals_the_weighted_bucket_sum(c, chunk, FG):
    rng = random.Random(c * 100 + chunk + FG)
    nb = 1 << (c - 1)
    chunk = min(chunk, nb)
    nwin = 3
    buckets = [[rng.randrange(P) for _ in range(nb)] for _ in range(nwin)]
    want = direct(buckets, nwin, c, nb)
    assert split_tail(buckets, nwin, c, nb, chunk, FG) == want
    assert split_tail(buckets, nwin, c, nb, chunk, FG, fuse_ok=False) == want  # KYB_MSM_REDUCE=nofuse


def test_the_configs2_plan_has_104_chains_of_at_most_126_doublings():
    """2^20 BLS12-381 G1 points: 8 windows of 16 bits, 2^15 buckets, chunks of 8, fold groups of 64"""
    nwin, c, chunk, FG = 8, 16, 8, 64
    nchunks = (1 << 15) // chunk
    chbits, tz, LB = nchunks.bit_length() - 1, 3, 6
    n = nwin * (1 + chbits)
    d = [chain_doublings(b, nwin, c, FUSED_BITS, LB, tz, chbits) for b in range(n)]
    assert n == 104 and max(d) == 7 * 16 + 3 + 11 == 126
    assert sorted(d) == sorted([16 * w for w in range(8)] + [16 * w + 3 + k for w in range(8) for k in range(12)])


def test_tree_levels_never_ask_one_group_for_two_additions():
    """in reduce_coop_kernel's fused levels the bit tree (RUN slots) and the plain tree (TOT slots) share a coop_add per
    level: a group is active in at most one of them, reads a partner that is idle in the SAME slot at that level, and
    every group index stays inside the workgroup"""
    G = 16
    for k in range(FUSED_BITS):
        off = 1 << k
        t_adders, w_adders = [], []
        for gi in range(G):
            low = gi & (2 * off - 1)
            if (low & (low - 1)) == 0 and low < off:
                t_adders.append(gi)
            if low == 2 * off - 1:
                w_adders.append(gi)
        assert not set(t_adders) & set(w_adders)
        assert all(0 <= g + off < G and g + off not in t_adders for g in t_adders)  # partner's RUN is not written this level
        assert all(0 <= g - off < G and g - off not in w_adders for g in w_adders)  # partner's TOT is not written this level
        assert len(t_adders) == (G >> (k + 1)) * (k + 1) and len(w_adders) == G >> (k + 1)
