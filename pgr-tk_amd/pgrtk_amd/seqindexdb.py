"""SeqIndexDB -- host-side mirror of pgr-tk's `pgrtk.SeqIndexDB` (pgr-tk/src/lib.rs:58-1404) for the
index / query subset.  Same method names, argument meaning and
defaults; the shimmer computation, the frag_map (as a sorted CSR) and the chaining run on the
GPU through libpgrhip.so.  Sequence iteration (FASTA) and the name tables stay on the host, as in
the reference.
"""
import ctypes as C
import gzip
import os

import numpy as np

from . import _ffi, mapgraph
from ._ffi import FRAG_REC, HITPAIR, HpsResult, default_context, lib
from .engine import frag_recs_batch, make_spec


def read_fastx(filepath):
    """FASTA / FASTQ (.gz ok) with the reference reader's record semantics, quirks included (pgr-db/src/fasta_io.rs:46-165;
    the same emulation as host/fastx.hpp): the first byte picks the format ('@' FASTQ, anything else FASTA) and is consumed;
    FASTA id = header up to the first space, the sequence runs to the next '>' ANYWHERE and drops '\\n' '>' '\\r'; FASTQ
    drops the record read last when the file ends right after its quality line (:159-162).  Returns [(name, seq bytes)]."""
    opener = gzip.open if filepath.endswith(".gz") else open
    with opener(filepath, "rb") as f:
        data = f.read()
    if not data:
        raise ValueError("empty file: " + filepath)
    n = len(data)
    pos = 1

    def read_until(delim):  # BufRead::read_until: bytes through the delimiter (included)
        nonlocal pos
        q = data.find(delim, pos) if pos < n else -1
        e = n if q < 0 else q + 1
        out = data[pos:e]
        pos = e
        return out

    def rec_id(head):
        return head.split(b" ", 1)[0].replace(b"\n", b"").replace(b"\r", b"").decode("utf-8", "replace")

    recs = []
    if data[:1] != b"@":
        while True:
            head = read_until(b"\n")
            if not head:
                break
            body = read_until(b">")
            recs.append((rec_id(head), body.replace(b"\n", b"").replace(b">", b"").replace(b"\r", b"")))
    else:
        while True:
            head = read_until(b"\n")
            seq = read_until(b"\n").replace(b"\n", b"").replace(b"\r", b"")
            read_until(b"+")
            read_until(b"\n")
            read_until(b"\n")
            if not read_until(b"@"):
                break
            recs.append((rec_id(head), seq))
    return recs


def _unpack_hps(res, n_queries):
    """pgr_hps_result -> per query: [(sid, [(score, [hitpair tuples])])]"""
    q_off = np.ctypeslib.as_array(res.q_off, shape=(n_queries + 1,)).copy()
    nt, nc, nh = int(res.n_targets), int(res.n_chains), int(res.n_hps)
    t_sid = np.ctypeslib.as_array(res.t_sid, shape=(max(nt, 1),))[:nt].copy()
    t_off = np.ctypeslib.as_array(res.t_off, shape=(nt + 1,)).copy()
    c_score = np.ctypeslib.as_array(res.c_score, shape=(max(nc, 1),))[:nc].copy()
    c_off = np.ctypeslib.as_array(res.c_off, shape=(nc + 1,)).copy()
    hps = np.zeros(nh, dtype=HITPAIR)
    if nh:
        C.memmove(hps.ctypes.data, res.hps, nh * HITPAIR.itemsize)
    # all hit pairs as Python tuples in one go (column lists zipped at C speed), then sliced per chain
    if nh:
        pairs = list(zip(zip(hps["qb"].tolist(), hps["qe"].tolist(), hps["qo"].tolist()),
                         zip(hps["tb"].tolist(), hps["te"].tolist(), hps["to"].tolist())))
    else:
        pairs = []
    q_off, t_off, c_off = q_off.tolist(), t_off.tolist(), c_off.tolist()
    t_sid, c_score = t_sid.tolist(), c_score.tolist()
    out = []
    for q in range(n_queries):
        targets = []
        for t in range(q_off[q], q_off[q + 1]):
            chains = [(c_score[c], pairs[c_off[c]:c_off[c + 1]]) for c in range(t_off[t], t_off[t + 1])]
            targets.append((t_sid[t], chains))
        out.append(targets)
    return out


def sparse_aln(sp_hits, max_span, penalty, max_gap=None, orientated=False, ctx=None):
    """pgrtk.sparse_aln (pgr-tk/src/lib.rs:1539-1550 -> aln::sparse_aln, aln.rs:12-142).
    sp_hits: [((qb,qe,qo),(tb,te,to))] -> [(score, [hit pairs])]"""
    ctx = ctx or default_context()
    a = np.array([(h[0][0], h[0][1], h[0][2], h[1][0], h[1][1], h[1][2]) for h in sp_hits], dtype=HITPAIR)
    if a.size < 2:
        raise ValueError("sparse_aln needs at least 2 hit pairs")  # aln.rs:24 assert
    off = (C.c_uint64 * 2)(0, a.size)
    res = HpsResult()
    ctx.check(lib().pgr_sparse_aln_batch(ctx.handle, 1, a.ctypes.data, off, max_span, penalty,
                                         int(max_gap is not None), int(max_gap or 0), int(bool(orientated)),
                                         C.byref(res)))
    out = _unpack_hps(res, 1)
    lib().pgr_hps_result_free(C.byref(res))
    return out[0][0][1] if out[0] else []


def sparse_aln_groups(groups, max_span, penalty, max_gap=None, orientated=False, ctx=None):
    """aln::sparse_aln on many independent groups of hit pairs in ONE call (pgr_sparse_aln_batch).
    -> {"chains": [[(score, [hit pairs])] per group], "n_nonterminating": groups the reference never finishes}"""
    ctx = ctx or default_context()
    flat = [h for g in groups for h in g]
    a = np.array([(h[0][0], h[0][1], h[0][2], h[1][0], h[1][1], h[1][2]) for h in flat], dtype=HITPAIR)
    off = np.zeros(len(groups) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(g) for g in groups])
    res = HpsResult()
    ctx.check(lib().pgr_sparse_aln_batch(ctx.handle, len(groups), a.ctypes.data, off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                         max_span, penalty, int(max_gap is not None), int(max_gap or 0),
                                         int(bool(orientated)), C.byref(res)))
    n_bad = int(res.n_nonterminating)
    out = _unpack_hps(res, 1)
    lib().pgr_hps_result_free(C.byref(res))
    by_group = {sid: chains for sid, chains in out[0]}
    return {"chains": [by_group.get(g, []) for g in range(len(groups))], "n_nonterminating": n_bad}


def get_shmmr_pairs_from_seq(seq, w=80, k=56, r=4, min_span=16, padding=False, ctx=None):
    """pgrtk.get_shmmr_pairs_from_seq (pgr-tk/src/lib.rs:1581-1613): [(h0, h1, p0, p1, orientation)]"""
    from .engine import sequence_to_shmmrs
    mm = sequence_to_shmmrs(0, seq, make_spec(w, k, r, min_span), padding=padding, ctx=ctx)
    out = []
    for a, b in zip(mm[:-1], mm[1:]):
        s0, s1 = int(a["x"]) >> 8, int(b["x"]) >> 8
        p0 = ((int(a["y"]) & 0xFFFFFFFF) >> 1) + 1
        p1 = ((int(b["y"]) & 0xFFFFFFFF) >> 1) + 1
        out.append((s0, s1, p0, p1, 0) if s0 < s1 else (s1, s0, p0, p1, 1))
    return out


def get_shmmr_dots(seq0, seq1, w=80, k=56, r=4, min_span=16, ctx=None):
    """lib.rs:1649-1697: shimmer matches between two sequences for a dot plot: (x, y) = positions in seq0 / seq1 of
    every pair of shimmers with the same hash, ordered by seq1 position, then by seq0 position"""
    from .engine import sequence_to_shmmrs_batch
    a, b = sequence_to_shmmrs_batch([seq0, seq1], make_spec(w, k, r, min_span, False), ctx=ctx or default_context())
    ha, pa = a["x"] >> np.uint64(8), ((a["y"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1)).astype(np.uint32)
    hb, pb = b["x"] >> np.uint64(8), ((b["y"] & np.uint64(0xFFFFFFFF)) >> np.uint64(1)).astype(np.uint32)
    by_hash = {}
    for h, p in zip(ha.tolist(), pa.tolist()):
        by_hash.setdefault(h, []).append(p)
    x, y = [], []
    for h, p in zip(hb.tolist(), pb.tolist()):
        for px in by_hash.get(h, ()):
            x.append(px)
            y.append(p)
    return x, y


def pgr_lib_version():
    """pgrtk.pgr_lib_version (pgr-tk/src/lib.rs:22-26: the build's VERSION_STRING): the version of libpgrhip.so"""
    return lib().pgr_version().decode()


class SeqIndexDB:
    """FASTX / MEMORY backends of the reference's SeqIndexDB (pgr-db/src/ext.rs:152-249)."""

    def __init__(self, ctx=None, device=0):
        self.ctx = ctx or default_context(device)
        self._ix = C.c_void_p()
        self._spec = None
        self.backend = "UNKNOWN"
        self.seq_index = None  # {(name, source): (sid, len)}   (lib.rs:215 getter)
        self.seq_info = None   # {sid: (name, source, len)}     (lib.rs:223 getter)
        self._n_seqs = 0
        self._host = None      # lazily downloaded sorted records + key table
        self._seqs = {}        # sid -> bytes (FASTX / MEMORY backends keep the sequences, ext.rs:344-489)

    # ------------------------------------------------------------------ loading
    def _reset(self, w, k, r, min_span):
        self.close()
        self._spec = make_spec(w, k, r, min_span, False)
        self._ix = C.c_void_p()
        self.ctx.check(lib().pgr_index_create(self.ctx.handle, C.byref(self._spec), C.byref(self._ix)))
        self.seq_index, self.seq_info, self._n_seqs, self._host, self._seqs = {}, {}, 0, None, {}

    def _append(self, named_seqs, source):
        seqs = [s for _, s in named_seqs]
        arrs, ptrs, lens, n = _ffi.seq_ptrs(seqs)
        sids = np.arange(self._n_seqs, self._n_seqs + n, dtype=np.uint32)
        self.ctx.check(lib().pgr_index_add_batch(self.ctx.handle, self._ix, n, ptrs, lens,
                                                 sids.ctypes.data_as(C.POINTER(C.c_uint32))))
        for i, (name, s) in enumerate(named_seqs):
            sid = self._n_seqs + i
            self.seq_index[(name, source)] = (sid, len(s))
            self.seq_info[sid] = (name, source, len(s))
            self._seqs[sid] = bytes(s)
        self._n_seqs += n
        self.ctx.check(lib().pgr_index_finalize(self.ctx.handle, self._ix))
        self._host = None

    def load_from_fastx(self, filepath, w=80, k=56, r=4, min_span=64):
        """lib.rs:142-155 / ext.rs:152-181"""
        self._reset(w, k, r, min_span)
        self.backend = "FASTX"
        self._append(read_fastx(filepath), filepath)

    def append_from_fastx(self, filepath):
        """lib.rs:157-165"""
        assert self.backend == "FASTX", "Only DB created with load_from_fastx() can add data from another fastx file"
        self._append(read_fastx(filepath), filepath)

    def load_from_seq_list(self, seq_list, source="Memory", w=80, k=56, r=4, min_span=8):
        """lib.rs:196-213 / ext.rs:208-249: seq_list = [(name, bytes)]"""
        self._reset(w, k, r, min_span)
        self.backend = "MEMORY"
        self._append([(n, bytes(s)) for n, s in seq_list], source)

    def load_from_mdb_index(self, prefix):
        """index-file backend (the reference's AGC / FRG backends load the same `.mdb` + `.midx` pair:
        ext.rs:87-150, seq_db.rs:1328-1471): the frag_map goes to the GPU, names come from the .midx.
        Fragment ids are the ones stored in the file."""
        self.close()
        self._ix = C.c_void_p()
        self.ctx.check(lib().pgr_index_load_mdb(self.ctx.handle, (prefix + ".mdb").encode(), C.byref(self._ix)))
        sp = _ffi.Spec()
        lib().pgr_index_spec(self._ix, C.byref(sp))
        self._spec = sp
        self.backend = "MDB"
        self.seq_index, self.seq_info, self._host = {}, {}, None
        with open(prefix + ".midx") as f:
            for line in f:
                sid, ln, name, source = line.rstrip("\n").split("\t")
                source = None if source == "-" else source
                self.seq_index[(name, source)] = (int(sid), int(ln))
                self.seq_info[int(sid)] = (name, source, int(ln))
        self._n_seqs = (max(self.seq_info) + 1) if self.seq_info else 0

    def close(self):
        if getattr(self, "_ix", None):
            if getattr(getattr(self, "ctx", None), "alive", True):  # (_ffi.Context.alive: not on an object of a destroyed context)
                lib().pgr_index_destroy(self._ix)
            self._ix = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ frag_map views
    def _records(self):
        """host copy of the CSR: (records sorted by key/sid/frg_id with GLOBAL fragment ids, {key: (lo, hi)})"""
        if self._host is None:
            p, n = C.c_void_p(), C.c_uint64()
            self.ctx.check(lib().pgr_index_download(self.ctx.handle, self._ix, C.byref(p), C.byref(n)))
            recs = _ffi.take(p, int(n.value), FRAG_REC)
            if self.backend in ("FASTX", "MEMORY"):
                # FASTX/MEMORY backends number fragments globally (seq_db.rs:189-357): per sequence
                # Prefix +1, one per pair, Suffix +1; a sequence without pairs takes 2 ids.
                pairs = np.bincount(recs["sid"], minlength=self._n_seqs).astype(np.int64)
                base = np.concatenate([[0], np.cumsum(np.where(pairs == 0, 2, pairs + 2))])[:-1]
                recs["frg_id"] = (base[recs["sid"]] + 1 + recs["frg_id"]).astype(np.uint32)
            keys = {}
            if len(recs):
                chg = np.flatnonzero((recs["h0"][1:] != recs["h0"][:-1]) | (recs["h1"][1:] != recs["h1"][:-1])) + 1
                starts = np.concatenate([[0], chg])
                ends = np.concatenate([chg, [len(recs)]])
                for s, e in zip(starts, ends):
                    keys[(int(recs["h0"][s]), int(recs["h1"][s]))] = (int(s), int(e))
            self._host = (recs, keys)
        return self._host

    @staticmethod
    def _sig(r):
        return (int(r["frg_id"]), int(r["sid"]), int(r["bgn"]), int(r["end"]), int(r["orient"]))

    def get_shmmr_spec(self):
        """lib.rs:729-735"""
        return None if self._spec is None else self._spec.as_tuple()

    def get_shmmr_map(self):
        """lib.rs:752-762: {(h0,h1): [(frg_id, sid, bgn, end, orientation)]}"""
        recs, keys = self._records()
        return {k: [self._sig(r) for r in recs[s:e]] for k, (s, e) in keys.items()}

    def get_shmmr_pair_list(self):
        """lib.rs:774-790: [(h0, h1, sid, bgn, end, orientation)] (order unspecified in the reference)"""
        recs, _ = self._records()
        return [(int(r["h0"]), int(r["h1"]), int(r["sid"]), int(r["bgn"]), int(r["end"]), int(r["orient"])) for r in recs]

    def get_shmmr_pair_count(self, shmmr_pair):
        """lib.rs:636-648"""
        _, keys = self._records()
        s, e = keys.get((int(shmmr_pair[0]), int(shmmr_pair[1])), (0, 0))
        return e - s

    def get_shmmr_pair_source_count(self, shmmr_pair, max_unique_count=None):
        """lib.rs:669-727: [(source name, count)] of the pair's fragment signatures per source file, sources with
        count >= max_unique_count dropped (order unspecified in the reference; sorted here)"""
        recs, keys = self._records()
        s, e = keys.get((int(shmmr_pair[0]), int(shmmr_pair[1])), (0, 0))
        count = {}
        for sid in recs["sid"][s:e].tolist():
            src = self.seq_info[sid][1] or ""
            count[src] = count.get(src, 0) + 1
        return sorted((k, v) for k, v in count.items() if max_unique_count is None or v < max_unique_count)

    # ------------------------------------------------------------------ queries
    def query_fragment(self, seq):
        """lib.rs:249-306 -> raw_query_fragment (seq_db.rs:1200-1228):
        [((h0,h1), (p0,p1,orientation), [fragment signatures])] in query order"""
        recs, keys = self._records()
        q = frag_recs_batch([seq], self._spec, query_side=True, ctx=self.ctx)[0]
        out = []
        for r in q:
            key = (int(r["h0"]), int(r["h1"]))
            s, e = keys.get(key, (0, 0))
            out.append((key, (int(r["bgn"]), int(r["end"]), int(r["orient"])), [self._sig(x) for x in recs[s:e]]))
        return out

    def get_match_positions_with_fragment(self, seq):
        """lib.rs:308-326 -> seq_db.rs:1271-1289: {sid: sorted [(bgn, end, direction)]}"""
        res = {}
        for _, (_, _, qo), sigs in self.query_fragment(seq):
            for (_, sid, p0, p1, d) in sigs:
                res.setdefault(sid, []).append((p0, p1, 0 if d == qo else 1))
        for v in res.values():
            v.sort()
        return res

    def query_fragments_to_hps(self, seqs, penalty, max_count=None, max_count_query=None, max_count_target=None,
                               max_aln_span=None, max_gap=None, orientated=False):
        """batched query_fragment_to_hps (the reference loops over queries with rayon,
        pgr-bin/src/bin/pgr-query.rs:135-165).  Defaults of the Option arguments: aln.rs:204-230."""
        arrs, ptrs, lens, n = _ffi.seq_ptrs(seqs)
        res = HpsResult()
        self.ctx.check(lib().pgr_query_hps_batch(
            self.ctx.handle, self._ix, n, ptrs, lens, float(penalty),
            128 if max_count is None else max_count, 128 if max_count_query is None else max_count_query,
            128 if max_count_target is None else max_count_target, 8 if max_aln_span is None else max_aln_span,
            int(max_gap is not None), int(max_gap or 0), int(bool(orientated)), C.byref(res)))
        out = _unpack_hps(res, n)
        lib().pgr_hps_result_free(C.byref(res))
        return out

    def query_fragment_to_hps(self, seq, penalty, max_count=None, max_count_query=None, max_count_target=None,
                              max_aln_span=None, max_gap=None, orientated=False):
        """lib.rs:365-420 -> ext.rs:252-282: [(sid, [(score, [((qb,qe,qo),(tb,te,to))])])]
        (target order: ascending sid; the reference's order is hash-map iteration order)"""
        return self.query_fragments_to_hps([seq], penalty, max_count, max_count_query, max_count_target, max_aln_span,
                                           max_gap, orientated)[0]

    def query_fragment_to_hps_from_mmap_file(self, seq, penalty, max_count=None, max_count_query=None,
                                             max_count_target=None, max_aln_span=None, max_gap=None, oriented=False):
        """ext.rs:285-342 (what pgr-query's default mode and pgr-web call): the reference answers from the mmap'ed `.mdb`
        through the `.midx` locations (raw_query_fragment_from_mmap_midx, seq_db.rs:1230-1269) and panics unless the backend is
        file based (AGC / FRG).  Here the `.mdb` of load_from_mdb_index lives on the GPU as the same CSR, so the call is the
        in-memory query on that backend -- and an error on the others, as in the reference."""
        if self.backend != "MDB":
            raise RuntimeError("the call query_fragment_to_hps_from_mmap_file() needs an index-file backend "
                               "(load_from_mdb_index); this database is " + str(self.backend))
        return self.query_fragment_to_hps(seq, penalty, max_count, max_count_query, max_count_target, max_aln_span, max_gap,
                                          oriented)

    # ------------------------------------------------------------------ sequences (lib.rs:809-890)
    def get_seq_by_id(self, sid):
        if sid not in self._seqs:
            raise KeyError("sequence %r is not held by this database (backend %s)" % (sid, self.backend))
        return self._seqs[sid]

    def get_seq(self, sample_name, ctg_name):
        return self.get_seq_by_id(self.seq_index[(ctg_name, sample_name)][0])

    def get_sub_seq_by_id(self, sid, bgn, end):
        return self.get_seq_by_id(sid)[bgn:end]

    def get_sub_seq(self, sample_name, ctg_name, bgn, end):
        return self.get_seq(sample_name, ctg_name)[bgn:end]

    # ------------------------------------------------------------------ MAP-graph / principal bundles (lib.rs:893-1300)
    def get_smp_adj_list(self, min_count, keeps=None):
        """lib.rs:893-919 -> seq_db::frag_map_to_adj_list: [(sid, (h0,h1,o), (h0,h1,o))]"""
        a = mapgraph.adj_list_records(self.ctx, self._ix, min_count, keeps)
        v, w = a["v"], a["w"]
        return list(zip(a["sid"].tolist(), zip(v["h0"].tolist(), v["h1"].tolist(), v["orient"].tolist()),
                        zip(w["h0"].tolist(), w["h1"].tolist(), w["orient"].tolist())))

    def sort_adj_list_by_weighted_dfs(self, adj_list, start):
        """lib.rs:938-985: [(node, parent, weight, is_leaf, global_rank, branch, branch_rank)]"""
        a = mapgraph.adj_records_from_tuples(self.ctx, self._ix, adj_list)
        return mapgraph.weighted_dfs(self.ctx, a, start, self.get_shmmr_pair_count((start[0], start[1])))

    def get_principal_bundles(self, min_count, path_len_cutoff, keeps=None):
        """lib.rs:1002-1013 -> ext.rs:491-510: [[(h0,h1,orientation)]] longest first"""
        return mapgraph.principal_bundles(self.ctx, self._ix, min_count, path_len_cutoff, keeps)

    def get_principal_bundle_decomposition(self, min_count, path_len_cutoff, keeps=None):
        """lib.rs:1066-1100: (principal_bundles [(id, mean order, [(h0,h1,dir)])],
        [(sid, [((h0,h1,p0,p1,o), (bundle id, direction, position) | None)])]) -- sequences in ascending sid
        (the reference's order is hash-map iteration order)"""
        bundles, by_sid = mapgraph.bundle_decomposition(self.ctx, self._ix, min_count, path_len_cutoff, keeps)
        return bundles, [(sid, by_sid.get(sid, [])) for sid in sorted(self.seq_info)]

    def get_principal_bundle_projection(self, min_count, path_len_cutoff, sequence, keeps=None):
        """lib.rs:1128-1146: like the decomposition for caller-provided [(sid, seq)]"""
        return mapgraph.bundle_projection(self.ctx, self._ix, min_count, path_len_cutoff, sequence, keeps)

    # ------------------------------------------------------------------ GFA / index text writers (ext.rs:652-960)
    def _gfa_lines(self, adj_list, vertex_map=None):
        """S / L lines of generate_mapg_gfa (ext.rs:727-788): segment ids in order of first appearance in the
        adjacency list, LN = mean pair span + k, SC = number of sequences supporting the link"""
        recs, keys = self._records()
        k = self._spec.k
        overlaps, frag_id = {}, {}
        for sid, v, w in adj_list:
            if v[0] <= w[0]:
                overlaps.setdefault((v, w), []).append((sid, v[2], w[2]))
                frag_id.setdefault((v[0], v[1]), len(frag_id))
                frag_id.setdefault((w[0], w[1]), len(frag_id))
        lines = ["H\tVN:Z:1.0\tCM:Z:Sparse Genome Graph Generated By pgr-tk"]
        for smp, i in frag_id.items():
            s, e = keys[smp]
            span = (recs["end"][s:e].astype(np.uint64) - recs["bgn"][s:e]).sum()
            ave_len = (int(span) & 0xFFFFFFFF) // (e - s)
            line = "S\t%d\t*\tLN:i:%d\tSN:Z:%016x_%016x" % (i, ave_len + k, smp[0], smp[1])
            if vertex_map is not None and smp in vertex_map:
                line += "\tBN:i:%d\tBP:i:%d" % (vertex_map[smp][0], vertex_map[smp][2])
            lines.append(line)
        for (v, w), vs in overlaps.items():
            lines.append("L\t%d\t%s\t%d\t%s\t%dM\tSC:i:%d" % (frag_id[(v[0], v[1])], "+" if v[2] == 0 else "-",
                                                              frag_id[(w[0], w[1])], "+" if w[2] == 0 else "-", k, len(vs)))
        return lines

    def _smp_adj_list_from_seqs(self, min_count, keeps):
        """seq_db::generate_smp_adj_list_for_seq (seq_db.rs:947-1002) for every sequence of the database"""
        recs, keys = self._records()
        order = np.lexsort((recs["frg_id"], recs["sid"]))
        keeps = set(keeps) if keeps is not None else None
        out = []
        prev = None
        for r in recs[order]:
            key = (int(r["h0"]), int(r["h1"]))
            o = int(r["orient"]) | int(key[0] == key[1])  # get_smps orientation (strict '<')
            cur = (int(r["sid"]), key, int(r["bgn"]), int(r["end"]), o, keys[key][1] - keys[key][0])
            if prev is not None and prev[0] == cur[0]:
                mc = 0 if (keeps is not None and cur[0] in keeps) else min_count
                if prev[5] >= mc and cur[5] >= mc and prev[3] == cur[2]:
                    out.append((cur[0], (prev[1][0], prev[1][1], prev[4]), (key[0], key[1], o)))
                    out.append((cur[0], (key[0], key[1], 1 - o), (prev[1][0], prev[1][1], 1 - prev[4])))
            prev = cur
        return out

    def generate_mapg_gfa(self, min_count, filepath, method="from_fragmap", keeps=None):
        """lib.rs:1304-1335 -> ext.rs:652-789 (line order inside the S and L blocks is hash-map order there)"""
        adj = self.get_smp_adj_list(min_count, keeps) if method == "from_fragmap" else \
            self._smp_adj_list_from_seqs(min_count, keeps)
        with open(filepath, "w") as f:
            f.write("\n".join(self._gfa_lines(adj)) + "\n")

    def generate_principal_mapg_gfa(self, min_count, path_len_cutoff, filepath, keeps=None):
        """lib.rs:1357-1384 -> ext.rs:849-960: the MAP-graph restricted to principal-bundle vertices, segments
        tagged with bundle id (BN) and position (BP)"""
        adj = self.get_smp_adj_list(min_count, keeps)
        pb = self.get_principal_bundles(min_count, path_len_cutoff, keeps) if adj else []
        vmap = {}
        for bid, path in enumerate(pb):
            for p, v in enumerate(path):
                vmap[(v[0], v[1])] = (bid, v[2], p)
        # filtered_adj_list (seq_db.rs:1097-1111): both ends on long DFS paths == both ends in some bundle
        filtered = [(sid, v, w) for sid, v, w in adj if (v[0], v[1]) in vmap and (w[0], w[1]) in vmap]
        with open(filepath, "w") as f:
            f.write("\n".join(self._gfa_lines(filtered, vmap)) + "\n")

    def write_mapg_idx(self, filepath):
        """ext.rs:791-847: K (spec), C (contigs), F (fragment signatures) lines"""
        recs, keys = self._records()
        w, k, r, ms, sk = self._spec.as_tuple()
        with open(filepath, "w") as f:
            f.write("K\t%d\t%d\t%d\t%d\t%s\n" % (w, k, r, ms, "true" if sk else "false"))
            for sid in sorted(self.seq_info):
                name, source, ln = self.seq_info[sid]
                f.write("C\t%d\t%s\t%s\t%d\n" % (sid, name, source if source is not None else "NA", ln))
            for (h0, h1), (s, e) in keys.items():
                for x in recs[s:e]:
                    f.write("F\t%016x_%016x\t%d\t%d\t%d\t%d\t%d\n" % (h0, h1, x["frg_id"], x["sid"], x["bgn"], x["end"],
                                                                      x["orient"]))

    def write_midx_to_text_file(self, filepath):
        """lib.rs:1337-1339 ("for backward compatibility"): the same file as write_mapg_idx"""
        self.write_mapg_idx(filepath)

    def write_frag_and_index_files(self, file_prefix):
        """lib.rs:1374-1384: for a database that holds its sequences (FASTX / MEMORY backends: `seq_db.is_some()`) the
        reference writes `<prefix>.sdx/.frg` (write_to_frag_files, the fragment-compressed sequences: SURVEY section 2 marks
        that store out of scope) and `<prefix>.mdb/.midx` (write_shmmr_map_index) -- the call
        `gen_frag_db.py` made to produce the golden `test_seqs_frag.mdb`.  The `.mdb/.midx` half is written here, with the
        backend's global fragment ids; on the other backends the call does nothing, as in the reference."""
        if self.backend in ("FASTX", "MEMORY"):
            self.write_shmmr_map_index(file_prefix)

    # ------------------------------------------------------------------ .mdb / .midx (seq_db.rs:790-810, 1291-1326)
    def write_shmmr_map_index(self, prefix):
        recs, keys = self._records()
        w, k, r, ms, sk = self._spec.as_tuple()
        with open(prefix + ".mdb", "wb") as f:
            f.write(b"mdb")
            f.write(np.array([w, k, r, ms, int(sk)], dtype="<u4").tobytes())
            f.write(np.array([len(keys)], dtype="<u8").tobytes())
            sig = np.dtype([("frg_id", "<u4"), ("sid", "<u4"), ("bgn", "<u4"), ("end", "<u4"), ("orient", "u1")])
            for (h0, h1), (s, e) in keys.items():
                f.write(np.array([h0, h1, e - s], dtype="<u8").tobytes())
                blk = np.zeros(e - s, dtype=sig)
                for name in ("frg_id", "sid", "bgn", "end"):
                    blk[name] = recs[name][s:e]
                blk["orient"] = recs["orient"][s:e]
                f.write(blk.tobytes())
        with open(prefix + ".midx", "w") as f:
            for sid in sorted(self.seq_info):  # an .midx may hold sparse sids (multi-file indexes, the sid quirk)
                name, source, ln = self.seq_info[sid]
                f.write("%d\t%d\t%s\t%s\n" % (sid, ln, name, source if source is not None else "-"))
