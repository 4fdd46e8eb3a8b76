"""MAP-graph / principal-bundle entry points of SeqIndexDB (pgr-tk/src/lib.rs:893-1300) on libpgrhip.

The adjacency list (sort + stencil over all frag_map records) and the bundle lookup of every shimmer pair
run on the GPU; the graph walks are serial host code inside the library (csrc/mapgraph.hip).
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import ADJ_PAIR, DFS_NODE, SMP_BUNDLE, VERTEX, Bundles, lib


def _keeps(keeps):
    if keeps is None:
        return None, 0, None
    a = np.ascontiguousarray(np.asarray(list(keeps), dtype=np.uint32))
    return a.ctypes.data_as(C.POINTER(C.c_uint32)), len(a), a


def _vtuple(v):
    return (int(v["h0"]), int(v["h1"]), int(v["orient"]))


def adj_list_records(ctx, ix, min_count, keeps=None):
    """seq_db::frag_map_to_adj_list -> numpy ADJ_PAIR array (vertex weights included)"""
    kp, nk, _hold = _keeps(keeps)
    p, n = C.c_void_p(), C.c_uint64()
    ctx.check(lib().pgr_index_adj_list(ctx.handle, ix, int(min_count), kp, nk, C.byref(p), C.byref(n)))
    return _ffi.take(p, int(n.value), ADJ_PAIR)


def adj_records_from_tuples(ctx, ix, adj_list):
    """[(sid, (h0,h1,o), (h0,h1,o))] -> ADJ_PAIR array with the weights frag_map[(h0,h1)].len() looked up on the GPU"""
    n = len(adj_list)
    a = np.zeros(n, dtype=ADJ_PAIR)
    if n == 0:
        return a
    a["sid"] = [t[0] for t in adj_list]
    for side, j in (("v", 1), ("w", 2)):
        a[side]["h0"] = [t[j][0] for t in adj_list]
        a[side]["h1"] = [t[j][1] for t in adj_list]
        a[side]["orient"] = [t[j][2] for t in adj_list]
    keys = np.empty((2 * n, 2), dtype=np.uint64)
    keys[:n, 0], keys[:n, 1] = a["v"]["h0"], a["v"]["h1"]
    keys[n:, 0], keys[n:, 1] = a["w"]["h0"], a["w"]["h1"]
    cnt = key_counts(ctx, ix, keys)
    a["v"]["count"], a["w"]["count"] = cnt[:n], cnt[n:]
    return a


def key_counts(ctx, ix, keys):
    keys = np.ascontiguousarray(keys, dtype=np.uint64).reshape(-1, 2)
    out = np.zeros(len(keys), dtype=np.uint32)
    ctx.check(lib().pgr_index_key_counts(ctx.handle, ix, len(keys), keys.ctypes.data, out.ctypes.data))
    return out


def weighted_dfs(ctx, adj, start, start_count):
    """seq_db::sort_adj_list_by_weighted_dfs: [(node, parent|None, weight, is_leaf, rank, branch, branch_rank)]"""
    sv = np.zeros(1, dtype=VERTEX)
    sv["h0"], sv["h1"], sv["orient"], sv["count"] = start[0], start[1], start[2], start_count
    adj = np.ascontiguousarray(adj)
    p, n = C.c_void_p(), C.c_uint64()
    ctx.check(lib().pgr_sort_adj_list_by_weighted_dfs(ctx.handle, adj.ctypes.data, len(adj), sv.ctypes.data, C.byref(p),
                                                      C.byref(n)))
    d = _ffi.take(p, int(n.value), DFS_NODE)
    return [(_vtuple(r["node"]), _vtuple(r["parent"]) if r["has_parent"] else None, int(r["node"]["count"]),
             bool(r["is_leaf"]), int(r["rank"]), int(r["branch"]), int(r["branch_rank"])) for r in d]


def _unpack_bundles(b, with_id):
    nb = int(b.n_bundles)
    off = np.ctypeslib.as_array(b.b_off, shape=(nb + 1,)).copy() if nb else np.zeros(1, dtype=np.uint64)
    verts = np.zeros(int(b.n_vertices), dtype=VERTEX)
    if b.n_vertices:
        C.memmove(verts.ctypes.data, b.vertices, verts.nbytes)
    out = []
    for i in range(nb):
        seg = verts[int(off[i]):int(off[i + 1])]
        path = list(zip(seg["h0"].tolist(), seg["h1"].tolist(), seg["orient"].tolist()))
        out.append((int(b.bundle_id[i]), int(b.mean_ord[i]), path) if with_id else path)
    return out


def principal_bundles(ctx, ix, min_count, path_len_cutoff, keeps=None):
    kp, nk, _hold = _keeps(keeps)
    b = Bundles()
    ctx.check(lib().pgr_principal_bundles(ctx.handle, ix, int(min_count), int(path_len_cutoff), kp, nk, C.byref(b)))
    out = _unpack_bundles(b, False)
    lib().pgr_bundles_free(C.byref(b))
    return out


def principal_bundles_from_adj(ctx, adj, path_len_cutoff):
    adj = np.ascontiguousarray(adj)
    b = Bundles()
    ctx.check(lib().pgr_principal_bundles_from_adj_list(ctx.handle, adj.ctypes.data, len(adj), int(path_len_cutoff),
                                                        C.byref(b)))
    out = _unpack_bundles(b, False)
    lib().pgr_bundles_free(C.byref(b))
    return out


def _smps_to_tuples(smps):
    """annotated shimmer pairs -> [((h0,h1,p0,p1,orientation), (bundle id, direction, position) | None)], columns
    converted and zipped at C speed"""
    if len(smps) == 0:
        return []
    pairs = zip(smps["h0"].tolist(), smps["h1"].tolist(), smps["bgn"].tolist(), smps["end"].tolist(), smps["orient"].tolist())
    infos = [None if b < 0 else (b, d, p) for b, d, p in
             zip(smps["bundle_id"].tolist(), smps["bundle_dir"].tolist(), smps["bundle_pos"].tolist())]
    return list(zip(pairs, infos))


def bundle_decomposition(ctx, ix, min_count, path_len_cutoff, keeps=None):
    """-> (principal_bundles_with_id, {sid: annotated smps})"""
    kp, nk, _hold = _keeps(keeps)
    b = Bundles()
    smps, n_smps, seq_sid, seq_off, n_seqs = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_void_p(), C.c_uint32()
    ctx.check(lib().pgr_principal_bundle_decomposition(ctx.handle, ix, int(min_count), int(path_len_cutoff), kp, nk,
                                                       C.byref(b), C.byref(smps), C.byref(n_smps), C.byref(seq_sid),
                                                       C.byref(seq_off), C.byref(n_seqs)))
    ns = int(n_seqs.value)
    bundles = _unpack_bundles(b, True)
    lib().pgr_bundles_free(C.byref(b))
    s = _ffi.take(smps, int(n_smps.value), SMP_BUNDLE)
    sid = _ffi.take(seq_sid, ns, np.dtype("<u4"))
    off = _ffi.take(seq_off, ns + 1, np.dtype("<u8"))
    return bundles, {int(sid[j]): _smps_to_tuples(s[int(off[j]):int(off[j + 1])]) for j in range(ns)}


def bundle_projection(ctx, ix, min_count, path_len_cutoff, sequences, keeps=None):
    """sequences = [(sid, seq)] -> (principal_bundles_with_id, [(sid, annotated smps)])"""
    kp, nk, _hold = _keeps(keeps)
    arrs, ptrs, lens, n = _ffi.seq_ptrs([s for _, s in sequences])
    sids = np.asarray([sid for sid, _ in sequences], dtype=np.uint32)
    b = Bundles()
    smps, n_smps, seq_off = C.c_void_p(), C.c_uint64(), C.c_void_p()
    ctx.check(lib().pgr_principal_bundle_projection(ctx.handle, ix, int(min_count), int(path_len_cutoff), kp, nk, n, ptrs,
                                                    lens, sids.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(b),
                                                    C.byref(smps), C.byref(n_smps), C.byref(seq_off)))
    bundles = _unpack_bundles(b, True)
    lib().pgr_bundles_free(C.byref(b))
    s = _ffi.take(smps, int(n_smps.value), SMP_BUNDLE)
    off = _ffi.take(seq_off, n + 1, np.dtype("<u8"))
    return bundles, [(int(sids[j]), _smps_to_tuples(s[int(off[j]):int(off[j + 1])])) for j in range(n)]
