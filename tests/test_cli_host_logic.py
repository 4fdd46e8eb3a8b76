"""CPU: host-side logic of the pgr-query counterpart (range merging, pgr-bin/src/bin/pgr-query.rs:167-285)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgr-tk_amd"))


def _hp(qb, tb, ln=100, qo=0, to=0):
    return ((qb, qb + ln, qo), (tb, tb + ln, to))


def test_chains_to_regions_merge_and_quirks():
    from pgrtk_amd.cli import chains_to_regions
    fwd1 = [_hp(0, 1000), _hp(200, 1200), _hp(400, 1400)]
    fwd2 = [_hp(600, 50000), _hp(800, 50200), _hp(1000, 50400)]
    far = [_hp(1200, 900000), _hp(1400, 900200), _hp(1600, 900400)]
    short = [_hp(0, 5), _hp(10, 15)]  # <= 2 hit pairs: dropped (pgr-query.rs:176)
    rev = [_hp(0, 7000, qo=0, to=1), _hp(200, 6800, qo=0, to=1), _hp(400, 6600, qo=0, to=1)]
    res = chains_to_regions([(3, [(10.0, fwd1), (9.0, fwd2), (8.0, far), (1.0, short)])], 100000)
    assert list(res) == [3]
    r = res[3]
    assert [(x[0], x[1], x[3], len(x[4])) for x in r] == [(1000, 50500, 0, 6), (900000, 900500, 0, 3)]
    # orientation vote uses running counters that are not reset per chain (quirk kept from the reference)
    res = chains_to_regions([(1, [(5.0, fwd1 + fwd1), (4.0, rev)])], 100000)
    assert [x[3] for x in res[1]] == [0]  # 6 forward vs 3 reverse after the second chain: still "forward"
    res = chains_to_regions([(1, [(4.0, rev), (5.0, fwd1)])], 100000)
    assert [(x[3], len(x[4])) for x in res[1]] == [(1, 6)]  # 3 reverse, then 3 vs 3 -> not f > r -> reverse again; merged
    assert chains_to_regions([(9, [(1.0, short)])], 100000) == {}


def test_reverse_complement():
    from pgrtk_amd.cli import reverse_complement
    assert reverse_complement(b"ACGTNacgt") == b"acgtNACGT"
