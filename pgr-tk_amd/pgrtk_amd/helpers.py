"""Python-level helpers of the reference package that sit on top of SeqIndexDB.query_fragment_to_hps
(pgr-tk/pgrtk/__init__.py: string_to_u8 / u8_to_string :93-127, query_sdb :130-221, merge_regions :270-328).
Host logic only; pinned on vectors generated from the reference's own functions (tests/golden/query_sdb_cases.json).
"""


# byte complement table (pgrtk/__init__.py:36-37) and the string one (:75), whose lower-case half pairs
# a->t, c->g, t->c, g->a in the reference: kept as is, the golden vectors pin it
_BYTE_RC = dict(zip(b"ACGTNnacgt", b"TGCANntgca"))
_STR_RC = dict(zip("ACGTNnactg", "TGCANntgca"))


def rc_byte_seq(seq):
    """reverse complement of a sequence given as a list of byte values"""
    return [_BYTE_RC[b] for b in reversed(seq)]


rc_u8_seq = rc_byte_seq


def rc(seq):
    """reverse complement of a sequence given as a str"""
    return "".join(_STR_RC[c] for c in reversed(seq))


def group_smps_by_principle_bundle_id(smps, len_cutoff=2500, merge_length=5000):
    """pgrtk/__init__.py:391-467 (the Python twin of pgr-pbundle-decomp.rs:61-137)"""
    from .cli import group_smps_by_principle_bundle_id as g
    return g(smps, len_cutoff, merge_length)


def string_to_u8(s):
    """DNA string -> list of byte values"""
    return list(s.encode("utf-8"))


def u8_to_string(u8):
    """list of byte values -> DNA string"""
    return bytes(u8).decode("utf-8")


def merge_regions(rgns, tol=1000):
    """rgns: [(bgn, end, length, orientation, aln_records)].  Sorted, then per orientation a region that starts less
    than `tol` after the END of the current merged region is folded into it (its end replaces the end, lengths and
    records add up); a region ending before the current end is dropped.  Returns forward regions, then reverse ones,
    as lists."""
    out = []
    ordered = sorted(rgns)
    for orientation in (0, 1):
        group = []
        for r in ordered:
            if r[3] != orientation:
                continue
            r = list(r)
            if not group:
                group.append(r)
                continue
            cur = group[-1]
            if r[1] < cur[1]:
                continue
            if r[0] - cur[1] < tol:
                cur[1] = r[1]
                cur[2] += r[2]
                cur[4] += r[4]
            else:
                group.append(r)
        out += group
    return out


def query_sdb(seq_index_db, query_seq, gap_penalty_factor=0.25, merge_range_tol=12, max_count=128, max_query_count=128,
              max_target_count=128, max_aln_span=8):
    """{target sid: [(start, end, length, orientation, hit pairs)]} for one query: chains with more than two hit
    pairs, orientation by the running forward / reverse vote of the target (the counters are not reset between the
    chains of a target, as in the reference), regions merged with merge_regions when merge_range_tol > 0."""
    res = seq_index_db.query_fragment_to_hps(query_seq, gap_penalty_factor, max_count, max_query_count, max_target_count,
                                             max_aln_span)
    ranges = {}
    for sid, chains in res:
        fwd = rev = 0
        for _score, aln in chains:
            if len(aln) <= 2:
                continue
            same = sum(1 for hp in aln if hp[0][2] == hp[1][2])
            fwd += same
            rev += len(aln) - same
            first = min((hp[1][0], hp[1][1]) for hp in aln)
            last = max((hp[1][0], hp[1][1]) for hp in aln)
            bgn, end = min(first), max(last)
            ranges.setdefault(sid, []).append((bgn, end, end - bgn, 0 if fwd > rev else 1, aln))
    if merge_range_tol > 0:
        for sid in ranges:
            ranges[sid] = merge_regions(ranges[sid], tol=merge_range_tol)
    return ranges


def get_principle_bundle_bed_file_for_query(seqs, w=64, k=56, r=4, min_span=32, min_cov=2, min_branch_length=8, ctx=None):
    """pgrtk/__init__.py:470-508: principal-bundle layout of a set of hit sequences whose names end in
    `_<bgn>_<end>_<direction>` (the names pgr-query / query_sdb give to fetched regions): [(ctg, bgn, end,
    "bundle:direction:first_pos:last_pos")] in contig-name order, partitions of a contig from last to first."""
    from .seqindexdb import SeqIndexDB
    sdb = SeqIndexDB(ctx=ctx)
    sdb.load_from_seq_list(seqs, "memory", w, k, r, min_span)
    _bundles, sid_smps = sdb.get_principal_bundle_decomposition(min_cov, min_branch_length)
    sid_smps = dict(sid_smps)
    layout = []
    for sid, (ctg, _src, _len) in sorted(sdb.seq_info.items(), key=lambda kv: kv[1][0]):
        ctg_bgn = int(ctg.split("_")[-3])
        for p in reversed(group_smps_by_principle_bundle_id(sid_smps[sid], 50, 100000)):
            layout.append((ctg, ctg_bgn + p[0][0][2], ctg_bgn + p[-1][0][3] + k,
                           "{}:{}:{}:{}".format(p[0][1], p[0][2], p[0][3], p[-1][3])))
    sdb.close()
    return layout
