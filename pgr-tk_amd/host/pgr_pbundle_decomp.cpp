// pgr-pbundle-decomp counterpart (pgr-bin/src/bin/pgr-pbundle-decomp.rs:26-531) in C++ above the C ABI:
//   pgr-pbundle-decomp <fastx> <out_prefix> [-w 48 -k 56 -r 4 --min-span 12 --min-cov 0 --min-branch-size 8
//                      --bundle-length-cutoff 2500 --bundle-merge-distance 10000 -d <decomp fastx> -i <include list>]
// MAP-graph principal bundles of the sequences of <fastx> (GPU: index, adjacency list, bundle lookup; library host
// code: the graph walks) and the bundle decomposition of every contig -> <out>.bed + <out>.ctg.summary.tsv.
// Not written: the .gfa / .mapg.idx / .pdb side files (the Python SeqIndexDB has the GFA / idx writers).
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "fastx.hpp"
#include "pgr_hip.h"

static void die(pgr_ctx *ctx, const char *what, int rc) {
    fprintf(stderr, "pgr-pbundle-decomp: %s failed (%d): %s\n", what, rc, pgr_last_error(ctx));
    exit(1);
}

struct Smp {  // ((h0,h1,p0,p1,orientation), Option<(bundle id, direction, position)>)
    uint64_t h0, h1;
    uint32_t bgn, end, orient;
    int32_t bid;
    uint32_t bdir, bpos;
};
struct Item {  // (smp, bid, direction, bpos) of group_smps_by_principle_bundle_id
    uint32_t bgn, end;
    uint32_t bid, d, bpos;
};

// rs:61-137
static std::vector<std::vector<Item>> group_smps(const std::vector<Smp> &smps, uint64_t cutoff, uint64_t merge_dist) {
    std::vector<std::vector<Item>> parts;
    std::vector<Item> cur;
    bool have = false;
    uint32_t pb = 0, pd = 0;
    auto long_enough = [&](const std::vector<Item> &p) { return (uint64_t)p.back().end - (uint64_t)p.front().bgn > cutoff; };
    for (const Smp &s : smps) {
        if (s.bid < 0) continue;
        const uint32_t d = s.orient == s.bdir ? 0u : 1u, bid = (uint32_t)s.bid;
        if (have && (bid != pb || d != pd)) {
            if (long_enough(cur)) parts.push_back(cur);
            cur.clear();
        }
        have = true;
        pb = bid;
        pd = d;
        cur.push_back(Item{s.bgn, s.end, bid, d, s.bpos});
    }
    if (!cur.empty() && long_enough(cur)) parts.push_back(cur);
    std::vector<std::vector<Item>> merged;
    for (auto &p : parts) {
        if (!merged.empty() && merged.back().back().bid == p.front().bid && merged.back().back().d == p.front().d &&
            (uint64_t)std::llabs((long long)p.front().bgn - (long long)merged.back().back().end) < merge_dist)
            merged.back().insert(merged.back().end(), p.begin(), p.end());
        else
            merged.push_back(std::move(p));
    }
    return merged;
}

static std::string f32s(float v) {  // Rust `{}` of an f32: shortest round-trip, no exponent
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
    char b[96];
    auto r = std::to_chars(b, b + sizeof b, v, std::chars_format::fixed);
    return std::string(b, r.ptr);
}

using pgrhost::with_extension;

int main(int argc, char **argv) {
    pgr_spec spec = {48, 56, 4, 12, 0};
    uint32_t min_cov = 0, min_branch = 8;
    uint64_t len_cutoff = 2500, merge_dist = 10000;
    std::string include, decomp_path;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * {
            if (i + 1 >= argc) {
                fprintf(stderr, "pgr-pbundle-decomp: %s needs a value\n", a.c_str());
                exit(2);
            }
            return argv[++i];
        };
        if (a == "-w") spec.w = (uint32_t)atoi(val());
        else if (a == "-k") spec.k = (uint32_t)atoi(val());
        else if (a == "-r") spec.r = (uint32_t)atoi(val());
        else if (a == "--min-span") spec.min_span = (uint32_t)atoi(val());
        else if (a == "--min-cov") min_cov = (uint32_t)atoi(val());
        else if (a == "--min-branch-size") min_branch = (uint32_t)atoi(val());
        else if (a == "--bundle-length-cutoff") len_cutoff = strtoull(val(), nullptr, 10);
        else if (a == "--bundle-merge-distance") merge_dist = strtoull(val(), nullptr, 10);
        else if (a == "-i" || a == "--include") include = val();
        else if (a == "-d" || a == "--decomp-fastx-path") decomp_path = val();
        else pos.push_back(a);
    }
    if (pos.size() != 2) {
        fprintf(stderr, "usage: pgr-pbundle-decomp <fastx_path> <output_prefix> [options]\n");
        return 2;
    }
    pgr_ctx *ctx = nullptr;
    int rc = pgr_ctx_create(0, &ctx);
    if (rc) {
        fprintf(stderr, "pgr-pbundle-decomp: pgr_ctx_create failed (%d): %s\n", rc, pgr_last_error(nullptr));
        return 1;
    }
    const std::vector<pgrhost::SeqRec> seqs = pgrhost::read_fastx(pos[0]);
    pgr_index *ix = nullptr;
    if ((rc = pgr_index_create(ctx, &spec, &ix))) die(ctx, "pgr_index_create", rc);
    {
        std::vector<const uint8_t *> ptrs;
        std::vector<uint64_t> lens;
        for (const auto &s : seqs) {
            ptrs.push_back((const uint8_t *)s.seq.data());
            lens.push_back(s.seq.size());
        }
        if ((rc = pgr_index_add_batch(ctx, ix, (uint32_t)seqs.size(), ptrs.data(), lens.data(), nullptr)))
            die(ctx, "pgr_index_add_batch", rc);
        if ((rc = pgr_index_finalize(ctx, ix))) die(ctx, "pgr_index_finalize", rc);
    }
    pgr_bundles bundles;
    pgr_smp_bundle *smps = nullptr;
    uint64_t n_smps = 0, *seq_off = nullptr;
    uint32_t *seq_sid = nullptr, n_seq = 0;
    if ((rc = pgr_principal_bundle_decomposition(ctx, ix, min_cov, min_branch, nullptr, 0, &bundles, &smps, &n_smps, &seq_sid,
                                                 &seq_off, &n_seq)))
        die(ctx, "pgr_principal_bundle_decomposition", rc);
    std::map<uint64_t, uint64_t> bundle_size;  // bundle id -> number of vertices
    for (uint64_t b = 0; b < bundles.n_bundles; ++b) bundle_size[bundles.bundle_id[b]] = bundles.b_off[b + 1] - bundles.b_off[b];

    struct Ctg {
        std::string name;
        uint64_t len;
        std::vector<Smp> smps;
    };
    std::vector<Ctg> ctgs;
    if (decomp_path.empty() && include.empty()) {
        ctgs.resize(seqs.size());
        for (size_t i = 0; i < seqs.size(); ++i) ctgs[i] = Ctg{seqs[i].name, seqs[i].seq.size(), {}};
        for (uint32_t j = 0; j < n_seq; ++j)
            for (uint64_t q = seq_off[j]; q < seq_off[j + 1]; ++q) {
                const pgr_smp_bundle &s = smps[q];
                ctgs[seq_sid[j]].smps.push_back(Smp{s.h0, s.h1, s.bgn, s.end, s.orient, s.bundle_id, s.bundle_dir, s.bundle_pos});
            }
    } else {
        // other sequences, annotated with the vertex map voted by <fastx_path>'s own sequences (rs:247-292,
        // ext.rs:976-1014): every bundle vertex is a shimmer pair of those sequences
        std::map<std::pair<uint64_t, uint64_t>, const pgr_smp_bundle *> vmap;
        for (uint64_t q = 0; q < n_smps; ++q)
            if (smps[q].bundle_id >= 0) vmap[{smps[q].h0, smps[q].h1}] = &smps[q];
        std::vector<pgrhost::SeqRec> other = pgrhost::read_fastx(decomp_path.empty() ? pos[0] : decomp_path);
        if (!include.empty()) {
            std::set<std::string> want;
            std::ifstream f(include);
            for (std::string l; std::getline(f, l);) {
                while (!l.empty() && (l.back() == '\r' || l.back() == ' ')) l.pop_back();
                if (!l.empty()) want.insert(l);
            }
            std::vector<pgrhost::SeqRec> keep;
            for (auto &r : other)
                if (want.count(r.name)) keep.push_back(std::move(r));
            other.swap(keep);
        }
        std::vector<const uint8_t *> ptrs;
        std::vector<uint64_t> lens;
        for (const auto &s : other) {
            ptrs.push_back((const uint8_t *)s.seq.data());
            lens.push_back(s.seq.size());
        }
        pgr_frag_rec *recs = nullptr;
        uint64_t *roff = nullptr;
        if ((rc = pgr_frag_recs_batch(ctx, &spec, (uint32_t)other.size(), ptrs.data(), lens.data(), nullptr, 1, &recs, &roff)))
            die(ctx, "pgr_frag_recs_batch", rc);
        for (size_t i = 0; i < other.size(); ++i) {
            Ctg c{other[i].name, other[i].seq.size(), {}};
            for (uint64_t q = roff[i]; q < roff[i + 1]; ++q) {
                const pgr_frag_rec &r = recs[q];
                Smp s{r.h0, r.h1, r.bgn, r.end, r.orient, -1, 0, 0};
                auto it = vmap.find({r.h0, r.h1});
                if (it != vmap.end()) {
                    s.bid = it->second->bundle_id;
                    s.bdir = it->second->bundle_dir;
                    s.bpos = it->second->bundle_pos;
                }
                c.smps.push_back(s);
            }
            ctgs.push_back(std::move(c));
        }
        pgr_free(recs);
        pgr_free(roff);
    }
    // contigs in name order (rs:343); ties keep the id order
    std::vector<size_t> order(ctgs.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return ctgs[a].name < ctgs[b].name; });

    FILE *bed = fopen(with_extension(pos[1], "bed").c_str(), "w"), *sum = fopen(with_extension(pos[1], "ctg.summary.tsv").c_str(), "w");
    if (!bed || !sum) {
        fprintf(stderr, "pgr-pbundle-decomp: can't write the outputs\n");
        return 1;
    }
    fputs("# cmd:", bed);
    for (int i = 0; i < argc; ++i) fprintf(bed, " %s", argv[i]);
    fputc('\n', bed);
    fputs("#ctg\tlength\trepeat_bundle_count\trepeat_bundle_sum\trepeat_bundle_percentage\trepeat_bundle_mean\t"
          "repeat_bundle_min\trepeat_bundle_max\tnon_repeat_bundle_count\tnon_repeat_bundle_sum\t"
          "non_repeat_bundle_percentage\tnon_repeat_bundle_mean\tnon_repeat_bundle_min\tnon_repeat_bundle_max\t"
          "total_bundle_count\ttotal_bundle_coverage_percentage\n",
          sum);
    std::vector<std::string> sum_lines;
    for (size_t oi : order) {
        const Ctg &c = ctgs[oi];
        const auto parts = group_smps(c.smps, len_cutoff, merge_dist);
        std::map<uint32_t, uint32_t> cnt;
        for (const auto &p : parts) ++cnt[p.front().bid];
        std::vector<uint32_t> rep, non;
        for (const auto &p : parts) {
            const uint32_t b = p.front().bgn, e = p.back().end + spec.k, bid = p.front().bid;
            const bool is_rep = cnt[bid] > 1;
            (is_rep ? rep : non).push_back(e - b - spec.k);
            fprintf(bed, "%s\t%u\t%u\t%u:%llu:%u:%u:%u:%s\n", c.name.c_str(), b, e, bid, (unsigned long long)bundle_size[bid],
                    p.front().d, p.front().bpos, p.back().bpos, is_rep ? "R" : "U");
        }
        auto stats = [&](const std::vector<uint32_t> &v, uint32_t total, std::string &mean, std::string &mn, std::string &mx) {
            if (v.empty()) {
                mean = mn = mx = "NA";
                return;
            }
            mean = f32s((float)total / (float)v.size());
            mn = std::to_string(*std::min_element(v.begin(), v.end()));
            mx = std::to_string(*std::max_element(v.begin(), v.end()));
        };
        uint32_t rs = 0, ns = 0;
        for (uint32_t v : rep) rs += v;
        for (uint32_t v : non) ns += v;
        std::string rmean, rmin, rmax, nmean, nmin, nmax;
        stats(rep, rs, rmean, rmin, rmax);
        stats(non, ns, nmean, nmin, nmax);
        const float len = (float)c.len;
        fprintf(sum, "%s\t%llu\t%zu\t%u\t%s\t%s\t%s\t%s\t%zu\t%u\t%s\t%s\t%s\t%s\t%zu\t%s\n", c.name.c_str(), (unsigned long long)c.len,
                rep.size(), rs, f32s(100.0f * (float)rs / len).c_str(), rmean.c_str(), rmin.c_str(), rmax.c_str(), non.size(), ns,
                f32s(100.0f * (float)ns / len).c_str(), nmean.c_str(), nmin.c_str(), nmax.c_str(), rep.size() + non.size(),
                f32s(100.0f * (float)(uint32_t)(rs + ns) / len).c_str());
    }
    fclose(bed);
    fclose(sum);
    fprintf(stderr, "%zu sequences, %llu principal bundles -> %s.bed / .ctg.summary.tsv\n", ctgs.size(),
            (unsigned long long)bundles.n_bundles, pos[1].c_str());
    pgr_bundles_free(&bundles);
    pgr_free(smps);
    pgr_free(seq_sid);
    pgr_free(seq_off);
    pgr_index_destroy(ix);
    pgr_ctx_destroy(ctx);
    return 0;
}
