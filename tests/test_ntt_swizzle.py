"""Host-side model of the register-tiled NTT's LDS tile exchanges (halo2-zkcert_amd/csrc/ntt.hip: place / swz / tile_rest and the two
kernels' group-0 ownerships): every ds access of every tile shape the planner can choose is replayed for each group of 32 lanes and must
hit 32 different banks (an element is 9 dwords, so the bank of its dword d is (9 a + d) mod 32: a permutation of the index's low five
bits).  The formulas below restate swz<SB, RULE>() and the host's choice of rule in ntt_run; the GPU side of the same claim is the absence of
k_ntt_* rows in the SQ_LDS_BANK_CONFLICT pass of profiles/*_pmc_*.csv."""
from collections import Counter

import pytest


def bitrev(v, bits):
    return int(format(v, f"0{bits}b")[::-1], 2) if bits else 0


def place(t, p, sb):          # thread index -> logical index with a zero sb-bit slot field at bit p
    return (t & ((1 << p) - 1)) | ((t >> p) << (p + sb))


def swz(L, z, sb):            # ntt.hip: swz<SB, RULE>()
    sh, mk, ts, rule_ = z
    if rule_ == 2:            # SWZ_REV (SB = 2): the row bits in reversed order
        x = L >> 5
        return L ^ bitrev(x & 15, 4) ^ (((x >> 5) & 1) << 3) ^ ((((x >> 1) ^ (x >> 4)) & 1) << 4)
    a = L ^ (((L >> sh) & mk) << ts)
    if rule_ == 1:            # SWZ_FIELD2: a second field, L[5 ...] of slot width, under the top of the bank bits
        a ^= ((L >> 5) & ((1 << sb) - 1)) << (5 - sb)
    return a


def rule(kind, sb, tlog, log_t):   # ntt.hip: ntt_run()
    if kind == "strided":
        return (log_t + sb, ((1 << (5 - log_t)) - 1) if log_t < 5 else 0, log_t, 0)
    if log_t + sb >= 5:
        return (tlog - 5, 31, 0, 0)
    if sb == 2 and log_t == 0:
        return (0, 0, 0, 2)
    return (tlog - 5, 31, 0, 1)


def worst_conflict(kind, sb, tlog, s):
    log_t = tlog - s
    z = rule(kind, sb, tlog, log_t)
    threads = (1 << tlog) >> sb
    worst, p_prev = 1, log_t
    for i in range(1, (s + sb - 1) // sb):
        e = min(sb * i + sb, s)
        p = log_t + e - sb
        for half in range(threads // 32):
            for q in range(1 << sb):
                wr, rd = [], []
                for lane in range(32):
                    t = half * 32 + lane
                    tf = t
                    if i == 1 and kind == "final":   # group 0 of the final pass: lanes run along j, rows are its bit reversal
                        jl, tl0 = t & ((1 << (s - sb)) - 1), t >> (s - sb)
                        tf = tl0 | (bitrev(jl, s - sb) << log_t)
                    wr.append(swz(place(tf, p_prev, sb) | (q << p_prev), z, sb))
                    rd.append(swz(place(t, p, sb) | (q << p), z, sb))
                for acc in (wr, rd):
                    assert len(set(acc)) == 32     # the map is a permutation of the tile
                    worst = max(worst, max(Counter((9 * a) % 32 for a in acc).values()))
        p_prev = p
    return worst


SHAPES = [(2, 10, s) for s in range(4, 11)] + [(2, 11, s) for s in range(4, 12)] + [(3, 11, s) for s in range(4, 12)]


@pytest.mark.parametrize("sb,tlog,s", SHAPES)
def test_tile_exchanges_are_bank_conflict_free(sb, tlog, s):
    """4 elements per thread on 1024- and 2048-element tiles, 8 per thread on 2048-element tiles; every pass width the planner produces"""
    assert worst_conflict("strided", sb, tlog, s) == 1
    assert worst_conflict("final", sb, tlog, s) == 1


def test_the_field_rule_alone_conflicts_on_narrow_final_tiles():
    """why the second rule exists: the final pass's 2^9 x 2 tile with the field rule is a 4-way conflict (what r03_v4's PMC pass counted)"""
    global rule
    keep = rule
    try:
        rule = lambda kind, sb, tlog, log_t: keep("strided", sb, tlog, log_t) if kind == "strided" else (tlog - 5, 31, 0, 0)   # noqa: E731
        assert worst_conflict("final", 2, 10, 9) == 4
    finally:
        rule = keep
