"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the MAP-graph / principal-bundle path that consumes the
frag_map (BASELINE.json configs[3]).

Pure-Python restatement, small cases only (config 4 = 96 x ~250 kbp runs in seconds).  Follows
  pgr-db/src/seq_db.rs:876-945    frag_map_to_adj_list
  pgr-db/src/seq_db.rs:1017-1062  sort_adj_list_by_weighted_dfs
  pgr-db/src/seq_db.rs:1064-1186  get_principal_bundles_from_adj_list
  pgr-db/src/graph_utils.rs:60-290  BiDiGraphWeightedDfs
  pgr-db/src/ext.rs:491-650       get_principal_bundles / get_principal_bundles_with_id
  pgr-db/src/ext.rs:976-1014      get_principal_bundle_decomposition
  pgr-bin/src/bin/pgr-pbundle-decomp.rs:61-137, 340-395  group_smps_by_principle_bundle_id + .bed lines

Third-party behaviour the reference relies on, restated from the published algorithms (the crates are
not vendored in /root/reference): petgraph 0.6.1 `GraphMap` (pgr-db/Cargo.toml:21; node and edge tables
are insertion-ordered IndexMaps, `remove_node` swap-removes), petgraph `Dfs`, and Rust std `BinaryHeap`
(sift-up on push; pop = swap root with last, sift down to the bottom, sift up) -- ties between equal
weights are resolved by these container mechanics, so they are modelled exactly.

PARITY UNPINNED: the reference holds no expected output for this path (no test, no fixture) and cannot
be run here (no Rust toolchain); this file pins the HIP/C++ product only oracle <-> product.
"""
import numpy as np

OUT, IN = 0, 1


class IndexMap:
    """insertion-ordered map with swap_remove (indexmap crate semantics)"""

    def __init__(self):
        self.keys = []
        self.vals = []
        self.pos = {}

    def __contains__(self, k):
        return k in self.pos

    def __len__(self):
        return len(self.keys)

    def get(self, k):
        i = self.pos.get(k)
        return None if i is None else self.vals[i]

    def insert(self, k, v):
        """-> True if the key was new"""
        i = self.pos.get(k)
        if i is None:
            self.pos[k] = len(self.keys)
            self.keys.append(k)
            self.vals.append(v)
            return True
        self.vals[i] = v
        return False

    def entry_or_insert(self, k, mk):
        i = self.pos.get(k)
        if i is None:
            self.pos[k] = len(self.keys)
            self.keys.append(k)
            self.vals.append(mk())
            return self.vals[-1]
        return self.vals[i]

    def swap_remove(self, k):
        i = self.pos.pop(k, None)
        if i is None:
            return None
        v = self.vals[i]
        lk, lv = self.keys.pop(), self.vals.pop()
        if i < len(self.keys):
            self.keys[i], self.vals[i] = lk, lv
            self.pos[lk] = i
        return v

    def copy(self):
        c = IndexMap()
        c.keys = list(self.keys)
        c.vals = [list(v) if isinstance(v, list) else v for v in self.vals]
        c.pos = dict(self.pos)
        return c


class DiGraphMap:
    """petgraph::graphmap::DiGraphMap<N, ()> (0.6.1)"""

    def __init__(self):
        self.nodes = IndexMap()   # N -> [(N, dir)]
        self.edges = IndexMap()   # (a, b) -> ()

    def add_edge(self, a, b):
        if self.edges.insert((a, b), None):
            self.nodes.entry_or_insert(a, list).append((b, OUT))
            if a != b:  # self loops have no Incoming entry
                self.nodes.entry_or_insert(b, list).append((a, IN))

    def node_list(self):
        return list(self.nodes.keys)

    def all_edges(self):
        return list(self.edges.keys)

    def neighbors_directed(self, a, d):
        lst = self.nodes.get(a)
        if lst is None:
            return []
        return [n for (n, dd) in lst if dd == d or n == a]

    def neighbors(self, a):
        lst = self.nodes.get(a)
        if lst is None:
            return []
        return [n for (n, dd) in lst if dd == OUT]

    def remove_node(self, n):
        links = self.nodes.swap_remove(n)
        if links is None:
            return False
        for succ, d in links:
            edge = (n, succ) if d == OUT else (succ, n)
            sus = self.nodes.get(succ)
            if sus is not None:
                want = (n, IN if d == OUT else OUT)
                for i, e in enumerate(sus):
                    if e == want:
                        last = sus.pop()
                        if i < len(sus):
                            sus[i] = last
                        break
            self.edges.swap_remove(edge)
        return True

    def clone(self):
        g = DiGraphMap()
        g.nodes = self.nodes.copy()
        g.edges = self.edges.copy()
        return g


class BinaryHeap:
    """Rust std::collections::BinaryHeap over (weight, payload), ordered by weight only"""

    def __init__(self):
        self.d = []

    def clear(self):
        self.d = []

    def is_empty(self):
        return not self.d

    def _sift_up(self, start, pos):
        d = self.d
        hole = d[pos]
        while pos > start:
            parent = (pos - 1) // 2
            if hole[0] <= d[parent][0]:
                break
            d[pos] = d[parent]
            pos = parent
        d[pos] = hole

    def push(self, item):
        self.d.append(item)
        self._sift_up(0, len(self.d) - 1)

    def pop(self):
        d = self.d
        item = d.pop()
        if d:
            item, d[0] = d[0], item
            end = len(d)
            pos = 0
            hole = d[0]
            child = 1
            while child <= max(end - 2, 0):  # end.saturating_sub(2)
                if d[child][0] <= d[child + 1][0]:
                    child += 1
                d[pos] = d[child]
                pos = child
                child = 2 * pos + 1
            if child == end - 1:
                d[pos] = d[child]
                pos = child
            d[pos] = hole
            self._sift_up(0, pos)
        return item


def rev(n):
    return (n[0], n[1], 1 - n[2])


def frag_map_to_adj_list(frag_map, min_count, keeps=None):
    """frag_map: {(h0,h1): [(frg_id, sid, bgn, end, orient), ...]}  -> [(sid, v, w)] (seq_db.rs:876-945)"""
    out = []
    for (h0, h1), sigs in frag_map.items():
        for (_f, sid, b, e, o) in sigs:
            out.append((sid, b, e, (h0, h1, o)))
    if len(out) < 2:
        return []
    out.sort()
    keeps = set(keeps) if keeps is not None else None
    kept = [(len(frag_map[(v[3][0], v[3][1])]) >= min_count) or (keeps is not None and v[0] in keeps) for v in out]
    adj = []
    for i in range(len(out) - 1):
        if kept[i] and kept[i + 1]:
            v, w = out[i], out[i + 1]
            if v[0] != w[0] or v[2] != w[1]:
                continue
            adj.append((v[0], v[3], w[3]))
            adj.append((v[0], rev(w[3]), rev(v[3])))
    return adj


def weighted_dfs(g, start, score):
    """BiDiGraphWeightedDfs::new + next() until None (graph_utils.rs:103-290)
    -> [(node, p_node, is_leaf, rank, branch, branch_rank)]"""
    pq = BinaryHeap()
    discovered = set()
    global_rank = {}
    s = score[start]
    pq.push((s, start))
    next_node = (s, start)
    global_rank[start] = 0
    current_branch = 0
    branch_rank_state = 0
    out = []
    while True:
        branch = current_branch
        res = None
        while True:
            if next_node is not None:
                node = next_node
                branch_rank = branch_rank_state
            else:
                if pq.is_empty():
                    return out
                node = pq.pop()
                branch_rank_state = 0
                branch_rank = 0
                current_branch += 1
                branch = current_branch
            n = node[1]
            if n in discovered:
                # the reference keeps looping with self.next_node unchanged; next_node can only be an
                # undiscovered node at this point, so this branch is reached from the heap only
                if next_node is not None:
                    raise RuntimeError("next_node already visited")
                continue
            discovered.add(n)
            rn = rev(n)
            discovered.add(rn)
            succ_f = []
            for succ in g.neighbors_directed(n, OUT):
                if n == succ or n == rev(succ):
                    continue
                if succ not in discovered:
                    succ_f.append((score[succ], succ))
            succ_r = []
            for succ in g.neighbors_directed(rn, OUT):
                if n == succ or n == rev(succ):
                    continue
                if succ not in discovered:
                    succ_r.append((score[succ], succ))
            is_leaf = False
            if not succ_f:
                is_leaf = True
                next_node = None
            if succ_f:
                succ_f.sort(key=lambda t: t[0])  # stable
                next_node = succ_f.pop()
                for t in succ_f:
                    pq.push(t)
            if succ_r:
                succ_r.sort(key=lambda t: t[0])
                for t in succ_r:
                    pq.push(t)
            node_rank = 0xFFFFFFFF
            p_node = None
            for m in g.neighbors_directed(n, IN):
                r = global_rank.get(m)
                if r is not None and r < node_rank:
                    node_rank, p_node = r, m
            for m in g.neighbors_directed(rn, IN):
                r = global_rank.get(m)
                if r is not None and r < node_rank:
                    node_rank, p_node = r, m
            if node_rank == 0xFFFFFFFF:
                node_rank = 0
            node_rank += 1
            global_rank[n] = node_rank
            global_rank[rn] = node_rank
            branch_rank_state += 1
            res = (n, p_node, is_leaf, node_rank, branch, branch_rank)
            break
        out.append(res)


def sort_adj_list_by_weighted_dfs(frag_map, adj_list, start):
    g = DiGraphMap()
    score = {}
    for (_sid, v, w) in adj_list:
        g.add_edge(v, w)
        if v not in score:
            score[v] = len(frag_map[(v[0], v[1])])
        if w not in score:
            score[w] = len(frag_map[(w[0], w[1])])
    return [(n, p, score[n], leaf, rank, br, brank) for (n, p, leaf, rank, br, brank) in weighted_dfs(g, start, score)]


def get_principal_bundles_from_adj_list(frag_map, adj_list, path_len_cutoff):
    """-> (principal_bundles, filtered_adj_list)   (seq_db.rs:1064-1186)"""
    assert adj_list
    s = adj_list[0][1]
    sorted_adj = sort_adj_list_by_weighted_dfs(frag_map, adj_list, s)
    paths, path = [], []
    for v in sorted_adj:
        path.append(v[0])
        if v[3]:
            paths.append(path)
            path = []
    main_vertices = set()
    for p in paths:
        if len(p) > path_len_cutoff:
            for v in p:
                main_vertices.add((v[0], v[1]))
    g0 = DiGraphMap()
    filtered = []
    for (sid, v, w) in adj_list:
        if (v[0], v[1]) in main_vertices and (w[0], w[1]) in main_vertices:
            g0.add_edge(v, w)
            filtered.append((sid, v, w))
    g1 = g0.clone()
    terminal = set()
    for (v, w) in g0.all_edges():
        if len(g0.neighbors_directed(v, OUT)) > 1:
            terminal.add(v)
        if len(g0.neighbors_directed(w, IN)) > 1:
            terminal.add(v)
    starts = [v for v in g1.node_list() if len(g1.neighbors_directed(v, IN)) == 0]
    if not starts and len(g1.nodes):
        starts.append(g1.nodes.keys[0])
    bundles = []
    while starts:
        s = starts.pop()
        stack = [s]
        disc = set()
        path = []
        while stack:  # petgraph Dfs::next until a terminal vertex
            node = stack.pop()
            if node in disc:
                continue
            disc.add(node)
            for succ in g1.neighbors(node):
                if succ not in disc:
                    stack.append(succ)
            path.append(node)
            if node in terminal:
                break
        if path:
            for v in path:
                g1.remove_node(v)
                g1.remove_node(rev(v))
            starts = [v for v in g1.node_list() if len(g1.neighbors_directed(v, IN)) == 0]
            bundles.append(path)
        if not starts and len(g1.nodes):
            starts.append(g1.nodes.keys[0])
    bundles.sort(key=lambda p: -len(p))  # stable, descending length
    return bundles, filtered


def get_principal_bundles(frag_map, min_count, path_len_cutoff, keeps=None):
    adj = frag_map_to_adj_list(frag_map, min_count, keeps)
    if not adj:
        return []
    return get_principal_bundles_from_adj_list(frag_map, adj, path_len_cutoff)[0]


def vertex_map_from_bundles(pb):
    m = {}
    for bid, path in enumerate(pb):
        for p, v in enumerate(path):
            m[(v[0], v[1])] = (bid, v[2], p)
    return m


def get_principal_bundles_with_id(frag_map, seq_smps, min_count, path_len_cutoff, keeps=None):
    """seq_smps: [(sid, [(h0,h1,p0,p1,orient), ...])] (ext.rs get_smps: strict '<' orientation).
    -> (principal_bundles_with_id [(bid, mean_ord, bundle)], vertex_map)   (ext.rs:552-650)"""
    pb = get_principal_bundles(frag_map, min_count, path_len_cutoff, keeps)
    vmap = vertex_map_from_bundles(pb)
    orders = {}
    directions = {}
    for _sid, smps in seq_smps:
        visited = set()
        for order, v in enumerate(smps):
            b = vmap.get((v[0], v[1]))
            if b is None:
                continue
            if b[0] not in visited:
                orders.setdefault(b[0], []).append(order)
                visited.add(b[0])
            directions.setdefault(b[0], []).append(0 if b[1] == v[4] else 1)
    mod = []
    for bid in range(len(pb)):
        if bid in orders:
            o = orders[bid]
            ssum = np.float32(0.0)
            for x in o:
                ssum = np.float32(ssum + np.float32(x))
            mean_ord = int(np.float32(ssum / np.float32(len(o))))
            d = directions[bid]
            direction = 0 if sum(d) < (len(d) >> 1) else 1
            mod.append((mean_ord, bid, direction))
        else:
            mod.append((0xFFFFFFFFFFFFFFFF, bid, 0))
    mod.sort()
    with_id = []
    for (ordv, bid, direction) in mod:
        if direction == 1:
            rpb = [(v[0], v[1], 1 - v[2]) for v in reversed(pb[bid])]
            for p, v in enumerate(rpb):
                vmap[(v[0], v[1])] = (bid, v[2], p)
            bundle = rpb
        else:
            bundle = list(pb[bid])
        with_id.append((bid, ordv, bundle))
    return with_id, vmap


def get_principal_bundle_decomposition(vmap, seq_smps):
    return [(sid, [(v, vmap.get((v[0], v[1]))) for v in smps]) for sid, smps in seq_smps]


def group_smps_by_principle_bundle_id(smps, bundle_length_cutoff, bundle_merge_distance):
    """pgr-pbundle-decomp.rs:61-137"""
    pre_bid = pre_d = None
    all_parts, new_part = [], []
    for smp, info in smps:
        if info is None:
            continue
        d = 0 if smp[4] == info[1] else 1
        bid, bpos = info[0], info[2]
        if pre_bid is None:
            new_part = [(smp, bid, d, bpos)]
            pre_bid, pre_d = bid, d
            continue
        if bid != pre_bid or d != pre_d:
            if new_part[-1][0][3] - new_part[0][0][2] > bundle_length_cutoff:
                all_parts.append(new_part)
            new_part = []
            pre_bid, pre_d = bid, d
        new_part.append((smp, bid, d, bpos))
    if new_part and new_part[-1][0][3] - new_part[0][0][2] > bundle_length_cutoff:
        all_parts.append(new_part)
    if not all_parts:
        return []
    rtn = []
    part = list(all_parts[0])
    for p in all_parts[1:]:
        if part[-1][1] == p[0][1] and part[-1][2] == p[0][2] and abs(p[0][0][2] - part[-1][0][3]) < bundle_merge_distance:
            part.extend(p)
        else:
            rtn.append(part)
            part = list(p)
    if part:
        rtn.append(part)
    return rtn


def bed_lines(seq_names, decomposition, with_id, k, bundle_length_cutoff=2500, bundle_merge_distance=10000):
    """the .bed body of pgr-pbundle-decomp (rs:340-395): contigs in name order"""
    bid_to_size = {b[0]: len(b[2]) for b in with_id}
    sid_smps = dict(decomposition)
    lines = []
    for sid, name in sorted(seq_names.items(), key=lambda t: t[1]):
        parts = group_smps_by_principle_bundle_id(sid_smps[sid], bundle_length_cutoff, bundle_merge_distance)
        cnt = {}
        for p in parts:
            cnt[p[0][1]] = cnt.get(p[0][1], 0) + 1
        for p in parts:
            b = p[0][0][2]
            e = p[-1][0][3] + k
            bid = p[0][1]
            lines.append("%s\t%d\t%d\t%d:%d:%d:%d:%d:%s" % (name, b, e, bid, bid_to_size[bid], p[0][2], p[0][3], p[-1][3],
                                                            "R" if cnt[bid] > 1 else "U"))
    return lines


def gfa_lines(frag_map, adj_list, k, vertex_map=None):
    """generate_mapg_gfa / generate_principal_mapg_gfa bodies (ext.rs:727-788, 905-957)"""
    overlaps, frag_id = {}, {}
    for sid, v, w in adj_list:
        if v[0] <= w[0]:
            overlaps.setdefault((v, w), []).append((sid, v[2], w[2]))
            for n in (v, w):
                if (n[0], n[1]) not in frag_id:
                    frag_id[(n[0], n[1])] = len(frag_id)
    lines = ["H\tVN:Z:1.0\tCM:Z:Sparse Genome Graph Generated By pgr-tk"]
    for smp, i in frag_id.items():
        hits = frag_map[smp]
        ave_len = (sum(h[3] - h[2] for h in hits) & 0xFFFFFFFF) // len(hits)
        line = "S\t%d\t*\tLN:i:%d\tSN:Z:%016x_%016x" % (i, ave_len + k, smp[0], smp[1])
        if vertex_map is not None and smp in vertex_map:
            line += "\tBN:i:%d\tBP:i:%d" % (vertex_map[smp][0], vertex_map[smp][2])
        lines.append(line)
    for (v, w), vs in overlaps.items():
        lines.append("L\t%d\t%s\t%d\t%s\t%dM\tSC:i:%d" % (frag_id[(v[0], v[1])], "+" if v[2] == 0 else "-",
                                                          frag_id[(w[0], w[1])], "+" if w[2] == 0 else "-", k, len(vs)))
    return lines


def smp_adj_list_for_seq(smps, sid, frag_map, min_count):
    """generate_smp_adj_list_for_seq (seq_db.rs:947-1002); smps = get_smps of the sequence"""
    out = []
    for i in range(len(smps) - 1):
        v, w = smps[i], smps[i + 1]
        fv, fw = frag_map.get((v[0], v[1])), frag_map.get((w[0], w[1]))
        if fv is None or fw is None or len(fv) < min_count or len(fw) < min_count or v[3] != w[2]:
            continue
        out.append((sid, (v[0], v[1], v[4]), (w[0], w[1], w[4])))
        out.append((sid, (w[0], w[1], 1 - w[4]), (v[0], v[1], 1 - v[4])))
    return out


def _rust_f32(x):
    """Rust Display of an f32"""
    x = np.float32(x)
    if np.isnan(x):
        return "NaN"
    if np.isinf(x):
        return "inf" if x > 0 else "-inf"
    return np.format_float_positional(x, unique=True, trim="-")


def ctg_summary_lines(seq_info, decomposition, k, bundle_length_cutoff=2500, bundle_merge_distance=10000):
    """<prefix>.ctg.summary.tsv of pgr-pbundle-decomp (rs:340-530); seq_info = {sid: (name, source, len)}"""
    sid_smps = dict(decomposition)
    order = sorted(seq_info.items(), key=lambda t: t[1][0])
    repeat, non_repeat = {}, {}
    for sid, (_name, _src, _len) in order:
        parts = group_smps_by_principle_bundle_id(sid_smps[sid], bundle_length_cutoff, bundle_merge_distance)
        cnt = {}
        for p in parts:
            cnt[p[0][1]] = cnt.get(p[0][1], 0) + 1
        for p in parts:
            b = p[0][0][2]
            e = p[-1][0][3] + k
            (repeat if cnt[p[0][1]] > 1 else non_repeat).setdefault(sid, []).append(e - b - k)
    cols = ["ctg", "length", "repeat_bundle_count", "repeat_bundle_sum", "repeat_bundle_percentage", "repeat_bundle_mean",
            "repeat_bundle_min", "repeat_bundle_max", "non_repeat_bundle_count", "non_repeat_bundle_sum",
            "non_repeat_bundle_percentage", "non_repeat_bundle_mean", "non_repeat_bundle_min", "non_repeat_bundle_max",
            "total_bundle_count", "total_bundle_coverage_percentage"]
    lines = ["#" + "\t".join(cols)]
    f = np.float32
    for sid, (name, _src, ln) in order:
        r, n = repeat.get(sid, []), non_repeat.get(sid, [])
        rs, ns = sum(r), sum(n)
        row = [name, str(ln), str(len(r)), str(rs), _rust_f32(f(100.0) * f(rs) / f(ln)),
               _rust_f32(f(rs) / f(len(r))) if r else "NA", str(min(r)) if r else "NA", str(max(r)) if r else "NA",
               str(len(n)), str(ns), _rust_f32(f(100.0) * f(ns) / f(ln)),
               _rust_f32(f(ns) / f(len(n))) if n else "NA", str(min(n)) if n else "NA", str(max(n)) if n else "NA",
               str(len(r) + len(n)), _rust_f32(f(100.0) * f(rs + ns) / f(ln))]
        lines.append("\t".join(row))
    return lines
