// mapgraph.hip -- MAP-graph and principal bundles on top of the GPU-resident frag_map (SURVEY 8f rank 3,
// BASELINE.json configs[3]).
//
//   GPU : frag_map_to_adj_list (pgr-db/src/seq_db.rs:876-945): every record gets its key's multiplicity, one
//         LSD radix sort by (sid, bgn, end, h0, h1, orient) replaces the reference's par_sort, a 2-point stencil
//         + scan emits the edge pairs;  the bundle lookup of every shimmer pair of every sequence
//         (ext.rs:976-1014) is a binary search per pair in a sorted vertex table.
//   host: the graph walks themselves -- BiDiGraphWeightedDfs (graph_utils.rs:60-290),
//         get_principal_bundles_from_adj_list (seq_db.rs:1064-1186), the order / direction vote of
//         ext.rs:552-650 -- are small and serial in the reference too.  Their results depend on container
//         mechanics of third-party crates (petgraph 0.6.1 GraphMap = insertion-ordered IndexMaps with
//         swap_remove, petgraph Dfs, std BinaryHeap), which are modelled here on dense node ids.
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "pgr_index.h"
#include "pgr_device.h"

using namespace pgr;

namespace {

// ================================================================================================ device side
__global__ void key_count_kernel(const pgr_frag_rec *__restrict__ recs, uint64_t n, const uint64_t *__restrict__ key_off,
                                 uint64_t n_keys, pgr_frag_rec *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t lo = 0, hi = n_keys;  // largest k with key_off[k] <= i
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (key_off[mid] <= i) lo = mid;
        else hi = mid;
    }
    pgr_frag_rec r = recs[i];
    r._pad = (uint32_t)(key_off[lo + 1] - key_off[lo]);
    out[i] = r;
}

__device__ __forceinline__ bool in_sorted(const uint32_t *__restrict__ a, uint32_t n, uint32_t v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo < n && a[lo] == v;
}

// s: records in (sid, bgn, end, node) order, _pad = key multiplicity.  flags[i] = 1 iff (i, i+1) is an edge.
__global__ void adj_flag_kernel(const pgr_frag_rec *__restrict__ s, uint64_t n, uint32_t min_count,
                                const uint32_t *__restrict__ keeps, uint32_t n_keeps, int has_keeps,
                                uint32_t *__restrict__ flags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    uint32_t f = 0;
    if (i + 1 < n) {
        const pgr_frag_rec &v = s[i], &w = s[i + 1];
        const bool kv = v._pad >= min_count || (has_keeps && in_sorted(keeps, n_keeps, v.sid));
        const bool kw = w._pad >= min_count || (has_keeps && in_sorted(keeps, n_keeps, w.sid));
        f = (kv && kw && v.sid == w.sid && v.end == w.bgn) ? 1u : 0u;
    }
    flags[i] = f;
}

__global__ void adj_emit_kernel(const pgr_frag_rec *__restrict__ s, uint64_t n, const uint32_t *__restrict__ flags,
                                const uint64_t *__restrict__ rank, pgr_adj_pair *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 >= n || !flags[i]) return;
    const pgr_frag_rec &v = s[i], &w = s[i + 1];
    pgr_adj_pair a, b;
    a.sid = b.sid = v.sid;
    a._pad = b._pad = 0;
    a.v = pgr_vertex{v.h0, v.h1, v.orient, v._pad};
    a.w = pgr_vertex{w.h0, w.h1, w.orient, w._pad};
    b.v = pgr_vertex{w.h0, w.h1, 1u - w.orient, w._pad};
    b.w = pgr_vertex{v.h0, v.h1, 1u - v.orient, v._pad};
    out[2 * rank[i]] = a;
    out[2 * rank[i] + 1] = b;
}

__global__ void key_counts_kernel(const uint64_t *__restrict__ keys, uint64_t nq, const pgr_frag_rec *__restrict__ recs,
                                  const uint64_t *__restrict__ key_off, uint64_t n_keys, uint32_t *__restrict__ counts) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nq) return;
    const uint64_t h0 = keys[2 * p], h1 = keys[2 * p + 1];
    uint64_t lo = 0, hi = n_keys;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        const pgr_frag_rec &r = recs[key_off[mid]];
        if (r.h0 < h0 || (r.h0 == h0 && r.h1 < h1)) lo = mid + 1;
        else hi = mid;
    }
    uint32_t c = 0;
    if (lo < n_keys) {
        const pgr_frag_rec &r = recs[key_off[lo]];
        if (r.h0 == h0 && r.h1 == h1) c = (uint32_t)(key_off[lo + 1] - key_off[lo]);
    }
    counts[p] = c;
}

// sorted vertex table entry: key -> (bundle id, direction, position)
struct VEntry {
    uint64_t h0, h1;
    int32_t bid;
    uint32_t dir, pos, _pad;
};

// every shimmer pair (a pair record in sequence order) -> its bundle annotation
__global__ void bundle_lookup_kernel(const pgr_frag_rec *__restrict__ recs, uint64_t n, int index_side,
                                     const VEntry *__restrict__ tab, uint64_t n_tab, pgr_smp_bundle *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const pgr_frag_rec r = recs[i];
    uint64_t lo = 0, hi = n_tab;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        const VEntry &e = tab[mid];
        if (e.h0 < r.h0 || (e.h0 == r.h0 && e.h1 < r.h1)) lo = mid + 1;
        else hi = mid;
    }
    pgr_smp_bundle o;
    o.h0 = r.h0;
    o.h1 = r.h1;
    o.bgn = r.bgn;
    o.end = r.end;
    // get_smps (ext.rs:534-548) orients with the strict '<': equal hashes are orientation 1 there, 0 in the index
    o.orient = index_side ? (r.orient | (r.h0 == r.h1 ? 1u : 0u)) : r.orient;
    o.sid = r.sid;
    o.bundle_id = -1;
    o.bundle_dir = 0;
    o.bundle_pos = 0;
    o._pad = 0;
    if (lo < n_tab && tab[lo].h0 == r.h0 && tab[lo].h1 == r.h1) {
        o.bundle_id = tab[lo].bid;
        o.bundle_dir = tab[lo].dir;
        o.bundle_pos = tab[lo].pos;
    }
    out[i] = o;
}

// ================================================================================================ host side
struct K128 {
    uint64_t a, b;
    bool operator==(const K128 &o) const { return a == o.a && b == o.b; }
};
struct K128Hash {
    size_t operator()(const K128 &k) const {
        uint64_t x = k.a * 0x9E3779B97F4A7C15ull ^ (k.b + 0x7F4A7C15ull + (k.a << 6) + (k.a >> 2));
        x ^= x >> 29;
        return (size_t)(x * 0xBF58476D1CE4E5B9ull);
    }
};

// ShmmrGraphNode interning: node id = 2 * key id + orientation, reverse(node) = id ^ 1
struct NodeTable {
    std::unordered_map<K128, uint32_t, K128Hash> map;
    std::vector<K128> keys;
    std::vector<uint32_t> count;
    uint32_t key_id(uint64_t h0, uint64_t h1, uint32_t cnt) {
        auto it = map.find(K128{h0, h1});
        if (it != map.end()) return it->second;
        const uint32_t id = (uint32_t)keys.size();
        map.emplace(K128{h0, h1}, id);
        keys.push_back(K128{h0, h1});
        count.push_back(cnt);
        return id;
    }
    uint32_t node(const pgr_vertex &v) { return key_id(v.h0, v.h1, v.count) * 2 + (v.orient & 1u); }
    int64_t find_node(const pgr_vertex &v) const {
        auto it = map.find(K128{v.h0, v.h1});
        if (it == map.end()) return -1;
        return (int64_t)it->second * 2 + (v.orient & 1u);
    }
    pgr_vertex vertex(uint32_t node) const {
        const K128 &k = keys[node >> 1];
        return pgr_vertex{k.a, k.b, node & 1u, count[node >> 1]};
    }
    uint32_t n_nodes() const { return (uint32_t)keys.size() * 2; }
};

constexpr uint32_t DIR_OUT = 0, DIR_IN = 1;

// petgraph 0.6.1 DiGraphMap<N, ()>: `nodes` is an insertion-ordered IndexMap N -> Vec<(N, direction)>,
// `edges` an IndexMap (a, b) -> ().  Only the node order and the per-node neighbour order are observable
// through the calls the reference makes; the edge table is kept as a set (it is only tested for membership).
struct GraphMap {
    std::vector<uint32_t> order;             // IndexMap key order
    std::vector<int32_t> pos;                // node -> index in order, -1 = absent
    std::vector<std::vector<uint32_t>> adj;  // node -> entries (neighbour * 2 + direction)
    std::unordered_set<uint64_t> edges;

    explicit GraphMap(uint32_t n_nodes) : pos(n_nodes, -1), adj(n_nodes) {}
    bool has(uint32_t n) const { return pos[n] >= 0; }
    void ensure(uint32_t n) {
        if (pos[n] < 0) {
            pos[n] = (int32_t)order.size();
            order.push_back(n);
        }
    }
    void add_edge(uint32_t a, uint32_t b) {
        if (!edges.insert(((uint64_t)a << 32) | b).second) return;
        ensure(a);
        adj[a].push_back(b * 2 + DIR_OUT);
        if (a != b) {  // self loops have no Incoming entry
            ensure(b);
            adj[b].push_back(a * 2 + DIR_IN);
        }
    }
    template <class F>
    void for_neighbors(uint32_t a, uint32_t dir, F &&f) const {
        if (!has(a)) return;
        for (uint32_t e : adj[a])
            if ((e & 1u) == dir || (e >> 1) == a) f(e >> 1);
    }
    uint32_t degree(uint32_t a, uint32_t dir) const {
        uint32_t c = 0;
        for_neighbors(a, dir, [&](uint32_t) { ++c; });
        return c;
    }
    void remove_node(uint32_t n) {
        if (!has(n)) return;
        const int32_t idx = pos[n];
        const uint32_t last = order.back();
        order.pop_back();
        if ((size_t)idx < order.size()) {
            order[idx] = last;
            pos[last] = idx;
        }
        pos[n] = -1;
        std::vector<uint32_t> links;
        links.swap(adj[n]);
        for (uint32_t e : links) {
            const uint32_t succ = e >> 1, dir = e & 1u;
            if (has(succ)) {
                std::vector<uint32_t> &sus = adj[succ];
                const uint32_t want = n * 2 + (dir ^ 1u);
                for (size_t i = 0; i < sus.size(); ++i)
                    if (sus[i] == want) {
                        sus[i] = sus.back();
                        sus.pop_back();
                        break;
                    }
            }
            edges.erase(dir == DIR_OUT ? (((uint64_t)n << 32) | succ) : (((uint64_t)succ << 32) | n));
        }
    }
};

// Rust std::collections::BinaryHeap<WeightedNode>: max-heap on the weight alone (graph_utils.rs:12-31)
struct WNode {
    uint32_t w, node;
};
struct BinHeap {
    std::vector<WNode> d;
    void sift_up(size_t start, size_t pos) {
        const WNode hole = d[pos];
        while (pos > start) {
            const size_t parent = (pos - 1) / 2;
            if (hole.w <= d[parent].w) break;
            d[pos] = d[parent];
            pos = parent;
        }
        d[pos] = hole;
    }
    void push(WNode x) {
        d.push_back(x);
        sift_up(0, d.size() - 1);
    }
    WNode pop() {
        WNode item = d.back();
        d.pop_back();
        if (!d.empty()) {
            std::swap(item, d[0]);
            const size_t end = d.size();
            size_t pos = 0;
            const WNode hole = d[0];
            size_t child = 1;
            const size_t lim = end >= 2 ? end - 2 : 0;
            while (child <= lim) {
                if (d[child].w <= d[child + 1].w) ++child;
                d[pos] = d[child];
                pos = child;
                child = 2 * pos + 1;
            }
            if (child == end - 1) {
                d[pos] = d[child];
                pos = child;
            }
            d[pos] = hole;
            sift_up(0, pos);
        }
        return item;
    }
};

struct DfsOut {
    uint32_t node;
    int64_t parent;
    bool is_leaf;
    uint32_t rank, branch, branch_rank;
};

// BiDiGraphWeightedDfs::new(g, start, score) + next() until None
void weighted_dfs(const GraphMap &g, const NodeTable &nt, uint32_t start, std::vector<DfsOut> &out) {
    const uint32_t nn = nt.n_nodes();
    std::vector<uint8_t> discovered(nn, 0);
    std::vector<uint32_t> global_rank(nn, 0);
    std::vector<uint8_t> has_rank(nn, 0);
    auto score = [&](uint32_t n) { return nt.count[n >> 1]; };
    BinHeap pq;
    pq.push(WNode{score(start), start});
    bool has_next = true;
    WNode next_node{score(start), start};
    global_rank[start] = 0;
    has_rank[start] = 1;
    uint32_t current_branch = 0, branch_rank_state = 0;
    std::vector<WNode> succ_f, succ_r;
    auto by_w = [](const WNode &a, const WNode &b) { return a.w < b.w; };
    for (;;) {
        uint32_t branch = current_branch, branch_rank;
        WNode node;
        if (has_next) {
            node = next_node;
            branch_rank = branch_rank_state;
        } else {
            if (pq.d.empty()) return;
            node = pq.pop();
            branch_rank_state = 0;
            branch_rank = 0;
            ++current_branch;
            branch = current_branch;
        }
        const uint32_t n = node.node, rn = n ^ 1u;
        if (discovered[n]) {
            if (has_next) return;  // cannot happen: next_node is always an undiscovered node
            continue;
        }
        discovered[n] = 1;
        discovered[rn] = 1;
        succ_f.clear();
        succ_r.clear();
        g.for_neighbors(n, DIR_OUT, [&](uint32_t s) {
            if (s == n || s == rn) return;  // do not walk through self loops
            if (!discovered[s]) succ_f.push_back(WNode{score(s), s});
        });
        g.for_neighbors(rn, DIR_OUT, [&](uint32_t s) {
            if (s == n || s == rn) return;
            if (!discovered[s]) succ_r.push_back(WNode{score(s), s});
        });
        bool is_leaf = false;
        if (succ_f.empty()) {
            is_leaf = true;
            has_next = false;
        } else {
            std::stable_sort(succ_f.begin(), succ_f.end(), by_w);
            next_node = succ_f.back();
            has_next = true;
            succ_f.pop_back();
            for (const WNode &s : succ_f) pq.push(s);
        }
        if (!succ_r.empty()) {
            std::stable_sort(succ_r.begin(), succ_r.end(), by_w);
            for (const WNode &s : succ_r) pq.push(s);
        }
        uint32_t node_rank = 0xFFFFFFFFu;
        int64_t p_node = -1;
        auto look = [&](uint32_t m) {
            if (has_rank[m] && global_rank[m] < node_rank) {
                node_rank = global_rank[m];
                p_node = m;
            }
        };
        g.for_neighbors(n, DIR_IN, look);
        g.for_neighbors(rn, DIR_IN, look);
        if (node_rank == 0xFFFFFFFFu) node_rank = 0;
        node_rank += 1;
        global_rank[n] = node_rank;
        has_rank[n] = 1;
        global_rank[rn] = node_rank;
        has_rank[rn] = 1;
        branch_rank_state += 1;
        out.push_back(DfsOut{n, p_node, is_leaf, node_rank, branch, branch_rank});
    }
}

struct AdjIds {
    std::vector<uint32_t> v, w;
};

void intern_adj(const pgr_adj_pair *adj, uint64_t n, NodeTable &nt, AdjIds &ids) {
    ids.v.resize(n);
    ids.w.resize(n);
    for (uint64_t i = 0; i < n; ++i) {
        ids.v[i] = nt.node(adj[i].v);
        ids.w[i] = nt.node(adj[i].w);
    }
}

// seq_db.rs:1064-1186
void principal_bundles_host(const pgr_adj_pair *adj, uint64_t n, uint32_t path_len_cutoff, NodeTable &nt,
                            std::vector<std::vector<uint32_t>> &bundles) {
    bundles.clear();
    if (n == 0) return;
    AdjIds ids;
    intern_adj(adj, n, nt, ids);
    const uint32_t nn = nt.n_nodes();
    std::vector<uint8_t> main_key(nt.keys.size(), 0);
    {
        GraphMap g(nn);
        for (uint64_t i = 0; i < n; ++i) g.add_edge(ids.v[i], ids.w[i]);
        std::vector<DfsOut> order;
        weighted_dfs(g, nt, ids.v[0], order);
        size_t path_start = 0;
        for (size_t i = 0; i < order.size(); ++i) {
            if (order[i].is_leaf) {
                const size_t len = i + 1 - path_start;
                if (len > path_len_cutoff)
                    for (size_t j = path_start; j <= i; ++j) main_key[order[j].node >> 1] = 1;
                path_start = i + 1;
            }
        }
    }
    GraphMap g0(nn);
    for (uint64_t i = 0; i < n; ++i)
        if (main_key[ids.v[i] >> 1] && main_key[ids.w[i] >> 1]) g0.add_edge(ids.v[i], ids.w[i]);
    std::vector<uint8_t> terminal(nn, 0);
    for (uint64_t e : g0.edges) {  // a set: iteration order is irrelevant
        const uint32_t v = (uint32_t)(e >> 32), w = (uint32_t)e;
        if (g0.degree(v, DIR_OUT) > 1) terminal[v] = 1;
        if (g0.degree(w, DIR_IN) > 1) terminal[v] = 1;
    }
    GraphMap g1 = g0;
    std::vector<uint32_t> starts;
    auto find_starts = [&]() {
        starts.clear();
        for (uint32_t v : g1.order)
            if (g1.degree(v, DIR_IN) == 0) starts.push_back(v);
    };
    find_starts();
    if (starts.empty() && !g1.order.empty()) starts.push_back(g1.order[0]);
    std::vector<uint32_t> stamp(nn, 0), stack;
    uint32_t cur = 0;
    while (!starts.empty()) {
        const uint32_t s = starts.back();
        starts.pop_back();
        ++cur;
        stack.clear();
        stack.push_back(s);
        std::vector<uint32_t> path;
        while (!stack.empty()) {  // petgraph Dfs::next until a terminal vertex
            const uint32_t node = stack.back();
            stack.pop_back();
            if (stamp[node] == cur) continue;
            stamp[node] = cur;
            if (g1.has(node))
                for (uint32_t e : g1.adj[node])
                    if ((e & 1u) == DIR_OUT && stamp[e >> 1] != cur) stack.push_back(e >> 1);
            path.push_back(node);
            if (terminal[node]) break;
        }
        if (!path.empty()) {
            for (uint32_t v : path) {
                g1.remove_node(v);
                g1.remove_node(v ^ 1u);
            }
            find_starts();
            bundles.push_back(std::move(path));
        }
        if (starts.empty() && !g1.order.empty()) starts.push_back(g1.order[0]);
    }
    std::stable_sort(bundles.begin(), bundles.end(),
                     [](const std::vector<uint32_t> &a, const std::vector<uint32_t> &b) { return a.size() > b.size(); });
}

void bundles_clear(pgr_bundles *b) { memset(b, 0, sizeof(*b)); }

// the host-only entry points (graph walks on caller-provided edges) work without a context / GPU
int host_fail(pgr_ctx *ctx, int code, const char *msg) { return ctx ? ctx->fail(code, msg) : code; }

int bundles_export(pgr_ctx *ctx, const NodeTable &nt, const std::vector<std::vector<uint32_t>> &bundles,
                   const std::vector<uint64_t> *ids, const std::vector<uint64_t> *ords, pgr_bundles *out) {
    bundles_clear(out);
    const size_t nb = bundles.size();
    size_t nv = 0;
    for (const auto &b : bundles) nv += b.size();
    out->b_off = (uint64_t *)malloc((nb + 1) * sizeof(uint64_t));
    out->bundle_id = (uint64_t *)malloc(std::max<size_t>(nb, 1) * sizeof(uint64_t));
    out->mean_ord = (uint64_t *)malloc(std::max<size_t>(nb, 1) * sizeof(uint64_t));
    out->vertices = (pgr_vertex *)malloc(std::max<size_t>(nv, 1) * sizeof(pgr_vertex));
    if (!out->b_off || !out->bundle_id || !out->mean_ord || !out->vertices) {
        pgr_bundles_free(out);
        return host_fail(ctx, PGR_ERR_NOMEM, "host allocation failed");
    }
    size_t o = 0;
    for (size_t b = 0; b < nb; ++b) {
        out->b_off[b] = o;
        out->bundle_id[b] = ids ? (*ids)[b] : b;
        out->mean_ord[b] = ords ? (*ords)[b] : 0;
        for (uint32_t v : bundles[b]) out->vertices[o++] = nt.vertex(v);
    }
    out->b_off[nb] = o;
    out->n_bundles = nb;
    out->n_vertices = nv;
    return PGR_OK;
}

int adj_list_device(pgr_ctx *ctx, const pgr_index *ix, uint32_t min_count, const uint32_t *keeps, uint32_t n_keeps,
                    pgr_adj_pair **out, uint64_t *n_out) {
    *out = nullptr;
    *n_out = 0;
    if (!ix->finalized) return ctx->fail(PGR_ERR_STATE, "index not finalized");
    PGR_ENTER(ctx);
    hipStream_t st = ctx->stream;
    const uint64_t n = ix->n;
    if (n < 2) return PGR_OK;
    int rc;
    Tmp withc(ctx), sorted(ctx), idx_a(ctx), idx_b(ctx), keys_a(ctx), keys_b(ctx), flags(ctx), rank(ctx), dkeeps(ctx);
    if ((rc = withc.alloc(n * sizeof(pgr_frag_rec))) || (rc = sorted.alloc(n * sizeof(pgr_frag_rec))) ||
        (rc = idx_a.alloc(n * 4)) || (rc = idx_b.alloc(n * 4)) || (rc = keys_a.alloc(n * 8)) || (rc = keys_b.alloc(n * 8)) ||
        (rc = flags.alloc((n + 1) * 4)) || (rc = rank.alloc((n + 1) * 8)))
        return rc;
    hipLaunchKernelGGL(key_count_kernel, grid_for(n), dim3(256), 0, st, ix->recs, n, ix->key_off, ix->n_keys,
                       withc.as<pgr_frag_rec>());
    launch_iota(st, idx_a.as<uint32_t>(), n);
    const int fields[6] = {F_ORIENT, F_H1, F_H0, F_END, F_BGN, F_SID};  // least significant first
    const unsigned bits[6] = {1, 56, 56, 32, 32, 32};
    if ((rc = sort_perm(ctx, withc.as<pgr_frag_rec>(), n, fields, bits, 6, idx_a.as<uint32_t>(), idx_b.as<uint32_t>(),
                        keys_a.as<uint64_t>(), keys_b.as<uint64_t>())))
        return rc;
    launch_gather_recs(st, withc.as<pgr_frag_rec>(), idx_a.as<uint32_t>(), sorted.as<pgr_frag_rec>(), n);
    std::vector<uint32_t> hk;
    if (keeps) {
        hk.assign(keeps, keeps + n_keeps);
        std::sort(hk.begin(), hk.end());
        if ((rc = dkeeps.alloc(std::max<size_t>(hk.size(), 1) * 4))) return rc;
        if (!hk.empty())
            PGR_HIP(ctx, hipMemcpyAsync(dkeeps.p, hk.data(), hk.size() * 4, hipMemcpyHostToDevice, st));
    }
    hipLaunchKernelGGL(adj_flag_kernel, grid_for(n + 1), dim3(256), 0, st, sorted.as<pgr_frag_rec>(), n, min_count,
                       dkeeps.as<uint32_t>(), (uint32_t)hk.size(), keeps ? 1 : 0, flags.as<uint32_t>());
    const size_t tb = scan_counts_temp_bytes((uint32_t)(n + 1));
    if ((rc = ctx->ws_scan_tmp.ensure(ctx, tb))) return rc;
    PGR_HIP(ctx, scan_counts(st, ctx->ws_scan_tmp.p, tb, flags.as<uint32_t>(), rank.as<uint64_t>(), (uint32_t)(n + 1)));
    uint64_t n_edges = 0;
    PGR_HIP(ctx, hipMemcpyAsync(&n_edges, rank.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, st));
    PGR_HIP(ctx, hipStreamSynchronize(st));
    if (n_edges == 0) return PGR_OK;
    Tmp dout(ctx);
    if ((rc = dout.alloc(2 * n_edges * sizeof(pgr_adj_pair)))) return rc;
    hipLaunchKernelGGL(adj_emit_kernel, grid_for(n), dim3(256), 0, st, sorted.as<pgr_frag_rec>(), n, flags.as<uint32_t>(),
                       rank.as<uint64_t>(), dout.as<pgr_adj_pair>());
    pgr_adj_pair *h = (pgr_adj_pair *)malloc(2 * n_edges * sizeof(pgr_adj_pair));
    if (!h) return ctx->fail(PGR_ERR_NOMEM, "host allocation failed");
    rc = ctx->d2h(h, dout.p, 2 * n_edges * sizeof(pgr_adj_pair));  // ordered after the kernel on the context's stream
    if (!rc && hipGetLastError() != hipSuccess) rc = ctx->fail(PGR_ERR_DEVICE, "adjacency kernels failed");
    if (rc) {
        free(h);
        return rc;
    }
    *out = h;
    *n_out = 2 * n_edges;
    return PGR_OK;
}

// vertex -> (bundle id, direction, position) of ext.rs:512-531 (later entries override earlier ones)
struct VMap {
    std::vector<int32_t> bid;   // per key id, -1 = none
    std::vector<uint32_t> dir, pos;
    explicit VMap(size_t n_keys) : bid(n_keys, -1), dir(n_keys, 0), pos(n_keys, 0) {}
    void set(uint32_t key, int32_t b, uint32_t d, uint32_t p) {
        bid[key] = b;
        dir[key] = d;
        pos[key] = p;
    }
};

int upload_vmap(pgr_ctx *ctx, const NodeTable &nt, const VMap &vm, Tmp &dtab, uint64_t *n_tab) {
    std::vector<VEntry> tab;
    for (size_t k = 0; k < vm.bid.size(); ++k)
        if (vm.bid[k] >= 0) tab.push_back(VEntry{nt.keys[k].a, nt.keys[k].b, vm.bid[k], vm.dir[k], vm.pos[k], 0});
    std::sort(tab.begin(), tab.end(),
              [](const VEntry &x, const VEntry &y) { return x.h0 < y.h0 || (x.h0 == y.h0 && x.h1 < y.h1); });
    *n_tab = tab.size();
    int rc;
    if ((rc = dtab.alloc(std::max<size_t>(tab.size(), 1) * sizeof(VEntry)))) return rc;
    // (ON the context's stream: a block from the caching allocator may still be read by work queued there -- and is filled there under
    // debug_poison --; a plain hipMemcpy runs on the null stream, which the context's non-blocking stream is not ordered with.  Found by
    // debug_poison: the fill landed on top of the table.)
    if (!tab.empty()) {
        PGR_HIP(ctx, hipMemcpyAsync(dtab.p, tab.data(), tab.size() * sizeof(VEntry), hipMemcpyHostToDevice, ctx->stream));
        PGR_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (`tab` goes out of scope)
    }
    return PGR_OK;
}

int lookup_smps(pgr_ctx *ctx, const pgr_frag_rec *d_recs, uint64_t n, int index_side, const NodeTable &nt,
                const VMap &vm, pgr_smp_bundle *d_out, pgr_smp_bundle *h_out) {
    Tmp dtab(ctx);
    uint64_t n_tab = 0;
    int rc;
    if ((rc = upload_vmap(ctx, nt, vm, dtab, &n_tab))) return rc;
    if (n == 0) return PGR_OK;
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(bundle_lookup_kernel, grid_for(n), dim3(256), 0, st, d_recs, n, index_side, dtab.as<VEntry>(), n_tab,
                       d_out);
    if ((rc = ctx->d2h(h_out, d_out, n * sizeof(pgr_smp_bundle)))) return rc;
    PGR_HIP(ctx, hipGetLastError());
    return PGR_OK;
}

// ext.rs:552-650 / pgr-tk/src/lib.rs:1147-1290 on smps held on the device in sequence order
// (seq_off: boundaries of the sequences inside d_recs, host)
int bundles_with_id(pgr_ctx *ctx, const pgr_index *ix, uint32_t min_count, uint32_t path_len_cutoff,
                    const uint32_t *keeps, uint32_t n_keeps, const pgr_frag_rec *d_recs, uint64_t n_smps, int index_side,
                    const std::vector<uint64_t> &seq_off, pgr_bundles *out_b, pgr_smp_bundle **out_smps) {
    bundles_clear(out_b);
    *out_smps = nullptr;
    pgr_adj_pair *adj = nullptr;
    uint64_t n_adj = 0;
    int rc = adj_list_device(ctx, ix, min_count, keeps, n_keeps, &adj, &n_adj);
    if (rc) return rc;
    NodeTable nt;
    std::vector<std::vector<uint32_t>> pb;
    principal_bundles_host(adj, n_adj, path_len_cutoff, nt, pb);
    free(adj);
    VMap vm(nt.keys.size());
    for (size_t b = 0; b < pb.size(); ++b)
        for (size_t p = 0; p < pb[b].size(); ++p) vm.set(pb[b][p] >> 1, (int32_t)b, pb[b][p] & 1u, (uint32_t)p);
    pgr_smp_bundle *h = (pgr_smp_bundle *)malloc(std::max<uint64_t>(n_smps, 1) * sizeof(pgr_smp_bundle));
    if (!h) return ctx->fail(PGR_ERR_NOMEM, "host allocation failed");
    Tmp dout(ctx);
    if ((rc = dout.alloc(std::max<uint64_t>(n_smps, 1) * sizeof(pgr_smp_bundle))) ||
        (rc = lookup_smps(ctx, d_recs, n_smps, index_side, nt, vm, dout.as<pgr_smp_bundle>(), h))) {
        free(h);
        return rc;
    }
    // vote: first position of every bundle along every sequence, and the direction of every hit
    const size_t nb = pb.size();
    std::vector<float> ord_sum(nb, 0.0f);
    std::vector<uint64_t> ord_n(nb, 0), dir_sum(nb, 0), dir_n(nb, 0);
    std::vector<uint64_t> seen(nb, ~0ull);
    for (size_t s = 0; s + 1 < seq_off.size(); ++s) {
        for (uint64_t i = seq_off[s]; i < seq_off[s + 1]; ++i) {
            const pgr_smp_bundle &v = h[i];
            if (v.bundle_id < 0) continue;
            const size_t b = (size_t)v.bundle_id;
            if (seen[b] != s) {
                seen[b] = s;
                ord_sum[b] += (float)(i - seq_off[s]);
                ord_n[b] += 1;
            }
            dir_sum[b] += (v.bundle_dir == v.orient) ? 0 : 1;
            dir_n[b] += 1;
        }
    }
    struct Mod {
        uint64_t mean_ord, bid;
        uint32_t direction;
    };
    std::vector<Mod> mod(nb);
    for (size_t b = 0; b < nb; ++b) {
        if (ord_n[b]) {
            const float mean = ord_sum[b] / (float)ord_n[b];
            mod[b] = Mod{(uint64_t)mean, b, dir_sum[b] < (dir_n[b] >> 1) ? 0u : 1u};
        } else {
            mod[b] = Mod{~0ull, b, 0u};
        }
    }
    std::sort(mod.begin(), mod.end(), [](const Mod &x, const Mod &y) {
        if (x.mean_ord != y.mean_ord) return x.mean_ord < y.mean_ord;
        if (x.bid != y.bid) return x.bid < y.bid;
        return x.direction < y.direction;
    });
    std::vector<std::vector<uint32_t>> with_id(nb);
    std::vector<uint64_t> ids(nb), ords(nb);
    for (size_t j = 0; j < nb; ++j) {
        const Mod &m = mod[j];
        ids[j] = m.bid;
        ords[j] = m.mean_ord;
        if (m.direction == 1) {
            const std::vector<uint32_t> &src = pb[m.bid];
            std::vector<uint32_t> r(src.size());
            for (size_t p = 0; p < src.size(); ++p) r[p] = src[src.size() - 1 - p] ^ 1u;
            for (size_t p = 0; p < r.size(); ++p) vm.set(r[p] >> 1, (int32_t)m.bid, r[p] & 1u, (uint32_t)p);  // override
            with_id[j] = std::move(r);
        } else {
            with_id[j] = pb[m.bid];
        }
    }
    if ((rc = lookup_smps(ctx, d_recs, n_smps, index_side, nt, vm, dout.as<pgr_smp_bundle>(), h)) ||
        (rc = bundles_export(ctx, nt, with_id, &ids, &ords, out_b))) {
        free(h);
        return rc;
    }
    *out_smps = h;
    return PGR_OK;
}

}  // namespace

// ================================================================================================ C ABI
extern "C" void pgr_bundles_free(pgr_bundles *b) {
    if (!b) return;
    free(b->b_off);
    free(b->bundle_id);
    free(b->mean_ord);
    free(b->vertices);
    memset(b, 0, sizeof(*b));
}

extern "C" int pgr_index_adj_list(pgr_ctx *ctx, const pgr_index *ix, uint32_t min_count, const uint32_t *keeps,
                                  uint32_t n_keeps, pgr_adj_pair **out, uint64_t *n_out) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!ix || !out || !n_out) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    return adj_list_device(ctx, ix, min_count, keeps, n_keeps, out, n_out);
}

extern "C" int pgr_index_key_counts(pgr_ctx *ctx, const pgr_index *ix, uint64_t n, const uint64_t *keys,
                                    uint32_t *counts) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!ix || (n && (!keys || !counts))) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    if (!ix->finalized) return ctx->fail(PGR_ERR_STATE, "index not finalized");
    if (n == 0) return PGR_OK;
    PGR_ENTER(ctx);
    hipStream_t st = ctx->stream;
    Tmp dk(ctx), dc(ctx);
    int rc;
    if ((rc = dk.alloc(n * 16)) || (rc = dc.alloc(n * 4))) return rc;
    PGR_HIP(ctx, hipMemcpyAsync(dk.p, keys, n * 16, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(key_counts_kernel, grid_for(n), dim3(256), 0, st, dk.as<uint64_t>(), n, ix->recs, ix->key_off,
                       ix->n_keys, dc.as<uint32_t>());
    PGR_HIP(ctx, hipMemcpyAsync(counts, dc.p, n * 4, hipMemcpyDeviceToHost, st));
    PGR_HIP(ctx, hipStreamSynchronize(st));
    PGR_HIP(ctx, hipGetLastError());
    return PGR_OK;
}

extern "C" int pgr_sort_adj_list_by_weighted_dfs(pgr_ctx *ctx, const pgr_adj_pair *adj, uint64_t n,
                                                 const pgr_vertex *start, pgr_dfs_node **out, uint64_t *n_out) {
    if (!out || !n_out || !start || (n && !adj)) return host_fail(ctx, PGR_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    *n_out = 0;
    NodeTable nt;
    AdjIds ids;
    intern_adj(adj, n, nt, ids);
    const int64_t s = nt.find_node(*start);
    if (s < 0) return host_fail(ctx, PGR_ERR_INVALID_ARG, "start node is not in the adjacency list");  // reference: expect()
    GraphMap g(nt.n_nodes());
    for (uint64_t i = 0; i < n; ++i) g.add_edge(ids.v[i], ids.w[i]);
    std::vector<DfsOut> order;
    weighted_dfs(g, nt, (uint32_t)s, order);
    pgr_dfs_node *h = (pgr_dfs_node *)malloc(std::max<size_t>(order.size(), 1) * sizeof(pgr_dfs_node));
    if (!h) return host_fail(ctx, PGR_ERR_NOMEM, "host allocation failed");
    for (size_t i = 0; i < order.size(); ++i) {
        pgr_dfs_node &o = h[i];
        memset(&o, 0, sizeof(o));
        o.node = nt.vertex(order[i].node);
        o.has_parent = order[i].parent >= 0;
        if (o.has_parent) o.parent = nt.vertex((uint32_t)order[i].parent);
        o.is_leaf = order[i].is_leaf;
        o.rank = order[i].rank;
        o.branch = order[i].branch;
        o.branch_rank = order[i].branch_rank;
    }
    *out = h;
    *n_out = order.size();
    return PGR_OK;
}

extern "C" int pgr_principal_bundles_from_adj_list(pgr_ctx *ctx, const pgr_adj_pair *adj, uint64_t n,
                                                   uint32_t path_len_cutoff, pgr_bundles *out) {
    if (!out || (n && !adj)) return host_fail(ctx, PGR_ERR_INVALID_ARG, "null argument");
    NodeTable nt;
    std::vector<std::vector<uint32_t>> pb;
    principal_bundles_host(adj, n, path_len_cutoff, nt, pb);
    return bundles_export(ctx, nt, pb, nullptr, nullptr, out);
}

extern "C" int pgr_principal_bundles(pgr_ctx *ctx, const pgr_index *ix, uint32_t min_count, uint32_t path_len_cutoff,
                                     const uint32_t *keeps, uint32_t n_keeps, pgr_bundles *out) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!ix || !out) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    pgr_adj_pair *adj = nullptr;
    uint64_t n = 0;
    int rc = adj_list_device(ctx, ix, min_count, keeps, n_keeps, &adj, &n);
    if (rc) return rc;
    rc = pgr_principal_bundles_from_adj_list(ctx, adj, n, path_len_cutoff, out);
    free(adj);
    return rc;
}

extern "C" int pgr_principal_bundle_decomposition(pgr_ctx *ctx, const pgr_index *ix, uint32_t min_count,
                                                  uint32_t path_len_cutoff, const uint32_t *keeps, uint32_t n_keeps,
                                                  pgr_bundles *bundles, pgr_smp_bundle **smps, uint64_t *n_smps,
                                                  uint32_t **seq_sid, uint64_t **seq_off, uint32_t *n_seqs) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!ix || !bundles || !smps || !n_smps || !seq_sid || !seq_off || !n_seqs)
        return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    if (!ix->finalized) return ctx->fail(PGR_ERR_STATE, "index not finalized");
    *smps = nullptr;
    *seq_sid = nullptr;
    *seq_off = nullptr;
    *n_smps = 0;
    *n_seqs = 0;
    bundles_clear(bundles);
    PGR_ENTER(ctx);
    hipStream_t st = ctx->stream;
    const uint64_t n = ix->n;
    int rc;
    // the shimmer pairs of every sequence in sequence order = the records sorted by (sid, frg_id)
    Tmp sorted(ctx), idx_a(ctx), idx_b(ctx), keys_a(ctx), keys_b(ctx);
    std::vector<uint64_t> off;
    std::vector<uint32_t> sids;
    if (n) {
        if ((rc = sorted.alloc(n * sizeof(pgr_frag_rec))) || (rc = idx_a.alloc(n * 4)) || (rc = idx_b.alloc(n * 4)) ||
            (rc = keys_a.alloc(n * 8)) || (rc = keys_b.alloc(n * 8)))
            return rc;
        launch_iota(st, idx_a.as<uint32_t>(), n);
        const int fields[2] = {F_FRG_ID, F_SID};
        const unsigned bits[2] = {32, 32};
        if ((rc = sort_perm(ctx, ix->recs, n, fields, bits, 2, idx_a.as<uint32_t>(), idx_b.as<uint32_t>(),
                            keys_a.as<uint64_t>(), keys_b.as<uint64_t>())))
            return rc;
        launch_gather_recs(st, ix->recs, idx_a.as<uint32_t>(), sorted.as<pgr_frag_rec>(), n);
        // sequence boundaries from the sorted sid column (keys_a is free again: reuse it for the sids)
        std::vector<pgr_frag_rec> hrec(n);
        PGR_HIP(ctx, hipMemcpyAsync(hrec.data(), sorted.p, n * sizeof(pgr_frag_rec), hipMemcpyDeviceToHost, st));
        PGR_HIP(ctx, hipStreamSynchronize(st));
        for (uint64_t i = 0; i < n; ++i)
            if (i == 0 || hrec[i].sid != hrec[i - 1].sid) {
                off.push_back(i);
                sids.push_back(hrec[i].sid);
            }
    }
    off.push_back(n);
    pgr_smp_bundle *h = nullptr;
    if ((rc = bundles_with_id(ctx, ix, min_count, path_len_cutoff, keeps, n_keeps, sorted.as<pgr_frag_rec>(), n, 1, off,
                              bundles, &h)))
        return rc;
    uint32_t *hs = (uint32_t *)malloc(std::max<size_t>(sids.size(), 1) * sizeof(uint32_t));
    uint64_t *ho = (uint64_t *)malloc(off.size() * sizeof(uint64_t));
    if (!hs || !ho) {
        free(hs);
        free(ho);
        free(h);
        pgr_bundles_free(bundles);
        return ctx->fail(PGR_ERR_NOMEM, "host allocation failed");
    }
    if (!sids.empty()) memcpy(hs, sids.data(), sids.size() * sizeof(uint32_t));
    memcpy(ho, off.data(), off.size() * sizeof(uint64_t));
    *smps = h;
    *n_smps = n;
    *seq_sid = hs;
    *seq_off = ho;
    *n_seqs = (uint32_t)sids.size();
    return PGR_OK;
}

extern "C" int pgr_principal_bundle_projection(pgr_ctx *ctx, const pgr_index *ix, uint32_t min_count,
                                               uint32_t path_len_cutoff, const uint32_t *keeps, uint32_t n_keeps,
                                               uint32_t n, const uint8_t *const *seqs, const uint64_t *lens,
                                               const uint32_t *sids, pgr_bundles *bundles, pgr_smp_bundle **smps,
                                               uint64_t *n_smps, uint64_t **seq_off) {
    if (!ctx) return PGR_ERR_INVALID_ARG;
    if (!ix || !bundles || !smps || !n_smps || !seq_off) return ctx->fail(PGR_ERR_INVALID_ARG, "null argument");
    if (!ix->finalized) return ctx->fail(PGR_ERR_STATE, "index not finalized");
    *smps = nullptr;
    *seq_off = nullptr;
    *n_smps = 0;
    bundles_clear(bundles);
    pgr_frag_rec *hrec = nullptr;
    uint64_t *hoff = nullptr;
    int rc = pgr_frag_recs_batch(ctx, &ix->spec, n, seqs, lens, sids, /*query_side=*/1, &hrec, &hoff);
    if (rc) return rc;
    const uint64_t np = hoff[n];
    std::vector<uint64_t> off(hoff, hoff + n + 1);
    Tmp drec(ctx);
    if ((rc = drec.alloc(std::max<uint64_t>(np, 1) * sizeof(pgr_frag_rec)))) {
        free(hrec);
        free(hoff);
        return rc;
    }
    hipError_t e = np ? hipMemcpyAsync(drec.p, hrec, np * sizeof(pgr_frag_rec), hipMemcpyHostToDevice, ctx->stream) : hipSuccess;  // (see upload_vmap)
    if (e == hipSuccess && np) e = hipStreamSynchronize(ctx->stream);
    free(hrec);
    if (e != hipSuccess) {
        free(hoff);
        return ctx->fail(PGR_ERR_DEVICE, hipGetErrorString(e));
    }
    pgr_smp_bundle *h = nullptr;
    rc = bundles_with_id(ctx, ix, min_count, path_len_cutoff, keeps, n_keeps, drec.as<pgr_frag_rec>(), np, 0, off, bundles,
                         &h);
    if (rc) {
        free(hoff);
        return rc;
    }
    *smps = h;
    *n_smps = np;
    *seq_off = hoff;
    return PGR_OK;
}
