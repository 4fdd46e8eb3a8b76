"""A SECOND, independent reading of the reference's sparse hit chaining -- test infrastructure only.

The chaining rows of SURVEY.md section 8 (a14-a16) have no expected output in the reference
(pgr-db/src/aln.rs:484 "TODO: Test the output properly") and the Rust cannot be compiled here, so the only
pin available is agreement between two restatements written separately.  `oracle/pgr_oracle.c` is the first
(arrays, indices, sorted runs); this file is the second: pure Python written from the Rust text alone, with
dicts / sets keyed by the hit-pair VALUE exactly as the reference's FxHashMap / FxHashSet are, and numpy
float32 scalars for every f32 operation in the reference's order.  tests/test_oracle_golden.py runs both on the
reference's `test_hits` fixture and on random groups and requires identical chains and bit-identical scores.

Follows: aln::sparse_aln                 pgr-db/src/aln.rs:12-142
         aln::query_fragment_to_hps      pgr-db/src/aln.rs:147-242
         seq_db::raw_query_fragment      pgr-db/src/seq_db.rs:1200-1228   (pair_shmmrs :102-111)

Where the reference's result depends on hash-set iteration order (aln.rs:112-118: ties between equal best
scores) the product's definition is used: among equal scores the hit pair with the lowest index in the stably
sorted hit list wins; the tests also compare the chains as canonically sorted sets.
"""
import numpy as np

f32 = np.float32


class WouldNotTerminate(Exception):
    """the reference loops forever on this input (aln.rs:105-131: no unvisited vertex has a positive score, or the
    predecessor map is cyclic)"""


def sparse_aln(sp_hits, max_span, penalty, max_gap=None, orientated=False):
    """sp_hits: list of ((qb, qe, qo), (tb, te, to)).  Returns [(score: np.float32, [hit pairs])] in extraction order."""
    penalty = f32(penalty)
    # :21  stable sort on the query begin only
    hits = sorted(sp_hits, key=lambda hp: hp[0][0])
    assert len(hits) > 1  # :24
    v_s = {}         # :22  score of every vertex, keyed by VALUE: identical hit pairs share an entry
    best_pre_v = {}  # :23
    first = hits[0]
    v_s[first] = f32(first[0][1]) - f32(first[0][0])  # :26
    best_pre_v[first] = None

    for i in range(1, len(hits)):  # :29
        hp = hits[i]
        best_v = None
        best_s = f32(0.0)
        span_set = set()
        j = i
        while j != 0:  # :35-39
            j -= 1
            pre = hits[j]
            if orientated:  # :43-50
                if (pre[0][2] ^ pre[1][2]) != (hp[0][2] ^ hp[1][2]):
                    continue
            if max_gap is not None:  # :52-65
                mg = f32(max_gap)
                dq = abs(f32(hp[0][0]) - f32(pre[0][1]))
                if hp[0][2] == hp[1][2]:
                    dt = abs(f32(hp[1][0]) - f32(pre[1][1]))
                else:
                    dt = abs(f32(hp[1][1]) - f32(pre[1][0]))
                if dq > mg or dt > mg:
                    continue
            if pre[0] == hp[0]:  # :67  same left coordinate
                continue
            span_set.add(pre[0])  # :70
            p_s = v_s.get(pre, f32(0.0))  # :71
            s = p_s + (f32(hp[0][1]) - f32(hp[0][0]))  # :72
            if hp[0][2] == hp[1][2]:  # :74-84
                s = s - penalty * (abs(f32(hp[0][0]) - f32(pre[0][1])) + abs(f32(hp[1][0]) - f32(pre[1][1])))
            else:
                s = s - penalty * (abs(f32(hp[0][0]) - f32(pre[0][1])) + abs(f32(hp[1][1]) - f32(pre[1][0])))
            if s > best_s:  # :86-89  strict: the first (nearest) of equal candidates stays
                best_s = s
                best_v = pre
            if len(span_set) >= max_span:  # :91
                break
        if best_s > f32(0.0):  # :96-102
            v_s[hp] = best_s
            best_pre_v[hp] = best_v
        else:
            v_s[hp] = f32(hp[0][1]) - f32(hp[0][0])
            best_pre_v[hp] = None

    # :105-140  extraction
    order = {}
    for idx, hp in enumerate(hits):
        order.setdefault(hp, idx)  # the product's definition of the unspecified FxHashSet order
    unvisited = set(hits)
    out = []
    # "the unvisited vertex with the highest score, ties to the smallest rank" over and over: scores no longer change,
    # so one ranking by (score descending, rank) visited front to back picks the same vertex as the reference's full scan
    # of the set (:112-118) does in every round
    ranked = sorted(unvisited, key=lambda hp: (-float(v_s.get(hp, f32(0.0))), order[hp]))
    nxt = 0
    while unvisited:
        while ranked[nxt] not in unvisited:
            nxt += 1
        best_v = ranked[nxt]
        best_s = v_s.get(best_v, f32(0.0))
        if not best_s > f32(0.0):  # :115 `*s > best_s` never fires: best_v stays None, :129-131 loops forever
            raise WouldNotTerminate()
        track = []
        in_walk = set()
        v = best_v
        while v is not None:  # :121-128
            if v not in unvisited or v in in_walk:
                # `v in in_walk`: a cycle in best_pre_v (only possible with duplicated hit pairs); the reference would
                # walk it forever, the product stops as if the vertex had been visited
                break
            track.append(v)
            in_walk.add(v)
            v = best_pre_v.get(v)
        if not track:  # :129-131 `continue` with nothing removed: an endless loop
            raise WouldNotTerminate()
        track.reverse()  # :132
        for hp in track:
            unvisited.discard(hp)
        bgn_s = v_s.get(track[0], f32(0.0))
        out.append((best_s - bgn_s, track))  # :138-139
    return out


def raw_query_fragment(frag_map, shmmrs):
    """seq_db.rs:1200-1228 given the query's shimmers [(hash, pos)] in order.
    frag_map: dict (h0, h1) -> [(frg_id, sid, bgn, end, orient)] in insertion order."""
    out = []
    for (s0, p0), (s1, p1) in zip(shmmrs[:-1], shmmrs[1:]):  # pair_shmmrs :102-111
        q0, q1 = p0 + 1, p1 + 1
        if s0 < s1:  # :1213 strict
            key, o = (s0, s1), 0
        else:
            key, o = (s1, s0), 1
        out.append((key, (q0, q1, o), list(frag_map.get(key, []))))
    return out


def query_fragment_to_hps(raw_query_hits, penalty, max_count=None, query_max_count=None, target_max_count=None,
                          max_aln_span=None, max_gap=None, oriented=False):
    """aln.rs:147-242 without the unused recomputation of the query's shimmers (:160-170 fills a map nobody reads).
    Returns {sid: [(score, [hit pairs])]} (the reference returns the same as a Vec in hash-map order)."""
    shmmr_pair_hash_count = {}
    target_shmer_pair_count = {}
    for key, _qpos, sigs in raw_query_hits:  # :172-193
        shmmr_pair_hash_count[key] = shmmr_pair_hash_count.get(key, 0) + 1
        for (_frg, sid, _b, _e, _o) in sigs:
            k3 = (key[0], key[1], sid)
            target_shmer_pair_count[k3] = target_shmer_pair_count.get(k3, 0) + 1
    by_sid = {}
    for key, qpos, sigs in raw_query_hits:  # :197-228
        count = shmmr_pair_hash_count.get(key, 0)
        if count > (128 if max_count is None else max_count):
            continue
        if count > (128 if query_max_count is None else query_max_count):
            continue
        for (_frg, sid, pos0, pos1, orient) in sigs:
            tcount = target_shmer_pair_count.get((key[0], key[1], sid), 0)
            if tcount > (128 if target_max_count is None else target_max_count):
                continue
            by_sid.setdefault(sid, []).append((qpos, (pos0, pos1, orient)))
    span = 8 if max_aln_span is None else max_aln_span  # :230
    return {sid: sparse_aln(hps, span, penalty, max_gap, oriented) for sid, hps in by_sid.items() if len(hps) > 1}
