// pgr-query counterpart (pgr-bin/src/bin/pgr-query.rs:17-409) in C++ above the C ABI of libpgrhip.so:
//   pgr-query <db> <query.fa> <out_prefix> [--fastx_file] [-w -k -r -m] [-g 0.025] [--merge-range-tol 100000]
//             [--max-count 128 --max-query-count 128 --max-target-count 128 --max-aln-chain-span 8]
//             [--only-summary] [--bed-summary]
// <db> is a <prefix> of .mdb/.midx files (default) or a FASTA file (--fastx_file).  The whole query batch goes to
// the GPU in one pgr_query_hps_batch call (the reference loops over queries with rayon, rs:135-138); chains ->
// per-target regions -> <out>.NNN.hit[.bed] (+ <out>.NNN.fa with --fastx_file unless --only-summary) is host code
// here as it is in the reference (rs:167-409), quirks included.
#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "fastx.hpp"
#include "pgr_hip.h"

static void die(pgr_ctx *ctx, const char *what, int rc) {
    fprintf(stderr, "pgr-query: %s failed (%d): %s\n", what, rc, pgr_last_error(ctx));
    exit(1);
}

struct HP {
    uint32_t qb, qe, qo, tb, te, to;
    bool operator<(const HP &o) const {
        return std::tie(qb, qe, qo, tb, te, to) < std::tie(o.qb, o.qe, o.qo, o.tb, o.te, o.to);
    }
};
struct Region {
    uint32_t bgn, end, len, orientation;
    std::vector<HP> aln;
    bool operator<(const Region &o) const {
        return std::tie(bgn, end, len, orientation, aln) < std::tie(o.bgn, o.end, o.len, o.orientation, o.aln);
    }
};
struct SeqInfo {
    std::string name, src;
    uint64_t len;
};

static std::string reverse_complement(const std::string &s) {  // pgr-db/src/fasta_io.rs:26-44
    std::string r(s.rbegin(), s.rend());
    for (char &c : r) switch (c) {
            case 'A': c = 'T'; break;
            case 'C': c = 'G'; break;
            case 'G': c = 'C'; break;
            case 'T': c = 'A'; break;
            case 'a': c = 't'; break;
            case 'c': c = 'g'; break;
            case 'g': c = 'c'; break;
            case 't': c = 'a'; break;
            default: break;
        }
    return r;
}

static std::string stem(const std::string &path) {  // basename without its last extension
    const size_t sl = path.find_last_of('/');
    std::string b = sl == std::string::npos ? path : path.substr(sl + 1);
    const size_t dot = b.find_last_of('.');
    if (dot != std::string::npos && dot > 0) b = b.substr(0, dot);
    return b;
}

using pgrhost::with_extension;

int main(int argc, char **argv) {
    pgr_spec spec = {80, 56, 4, 64, 0};
    float gap_penalty = 0.025f;
    uint32_t merge_tol = 100000, max_count = 128, max_q = 128, max_t = 128, max_span = 8;
    bool fastx_file = false, only_summary = false, bed_summary = false;
    size_t query_batch = 8192;  // more queries than this go to the GPU in batches of this size, two in flight (pgr_pipe_submit_query)
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * {
            if (i + 1 >= argc) {
                fprintf(stderr, "pgr-query: %s needs a value\n", a.c_str());
                exit(2);
            }
            return argv[++i];
        };
        if (a == "-w") spec.w = (uint32_t)atoi(val());
        else if (a == "-k") spec.k = (uint32_t)atoi(val());
        else if (a == "-r") spec.r = (uint32_t)atoi(val());
        else if (a == "-m" || a == "--min-span") spec.min_span = (uint32_t)atoi(val());
        else if (a == "-g" || a == "--gap-penalty-factor") gap_penalty = (float)atof(val());
        else if (a == "--merge-range-tol") merge_tol = (uint32_t)atoll(val());
        else if (a == "--max-count") max_count = (uint32_t)atoi(val());
        else if (a == "--max-query-count") max_q = (uint32_t)atoi(val());
        else if (a == "--max-target-count") max_t = (uint32_t)atoi(val());
        else if (a == "--max-aln-chain-span") max_span = (uint32_t)atoi(val());
        else if (a == "--query-batch") query_batch = (size_t)std::max(1ll, atoll(val()));
        // (clap 4 derives kebab-case long names from the struct fields, pgr-query.rs:26-66: --fastx-file, --frg-file, --only-summary ...;
        // the snake_case spellings of earlier rounds stay accepted)
        else if (a == "--fastx-file" || a == "--fastx_file") fastx_file = true;
        else if (a == "--frg-file" || a == "--frg_file") {
            fprintf(stderr, "pgr-query: --frg-file (the reference's .frg sequence store, pgr-query.rs:26-27) is not supported: this program reads "
                            "the .mdb/.midx pair (default) or, with --fastx-file, a FASTA/FASTQ(.gz) database\n");
            return 2;
        } else if (a == "--only-summary" || a == "--only_summary") only_summary = true;
        else if (a == "--bed-summary" || a == "--bed_summary") bed_summary = true;
        else if (a.size() > 1 && a[0] == '-' && !(a.size() > 1 && isdigit((unsigned char)a[1]))) {
            fprintf(stderr, "pgr-query: unknown option %s\n", a.c_str());
            return 2;
        } else pos.push_back(a);
    }
    if (pos.size() != 3) {
        fprintf(stderr, "usage: pgr-query <pgr_db_prefix | fasta> <query_fastx> <output_prefix> [--fastx-file] [-w 80 -k 56 -r 4 -m 64] [-g 0.025] [--merge-range-tol 100000] [--max-count 128 --max-query-count 128 --max-target-count 128 --max-aln-chain-span 8] [--only-summary] [--bed-summary] [--query-batch N]\n");
        return 2;
    }
    pgr_ctx *ctx = nullptr;
    int rc = pgr_ctx_create(0, &ctx);
    if (rc) {
        fprintf(stderr, "pgr-query: pgr_ctx_create failed (%d): %s\n", rc, pgr_last_error(nullptr));
        return 1;
    }
    // ---- database
    pgr_index *ix = nullptr;
    std::map<uint32_t, SeqInfo> seq_info;
    std::vector<pgrhost::SeqRec> db_seqs;
    if (fastx_file) {
        db_seqs = pgrhost::read_fastx(pos[0]);
        if ((rc = pgr_index_create(ctx, &spec, &ix))) die(ctx, "pgr_index_create", rc);
        std::vector<const uint8_t *> ptrs;
        std::vector<uint64_t> lens;
        for (size_t i = 0; i < db_seqs.size(); ++i) {
            ptrs.push_back((const uint8_t *)db_seqs[i].seq.data());
            lens.push_back(db_seqs[i].seq.size());
            seq_info[(uint32_t)i] = SeqInfo{db_seqs[i].name, pos[0], db_seqs[i].seq.size()};
        }
        if ((rc = pgr_index_add_batch(ctx, ix, (uint32_t)db_seqs.size(), ptrs.data(), lens.data(), nullptr)))
            die(ctx, "pgr_index_add_batch", rc);
        if ((rc = pgr_index_finalize(ctx, ix))) die(ctx, "pgr_index_finalize", rc);
    } else {
        if ((rc = pgr_index_load_mdb(ctx, (pos[0] + ".mdb").c_str(), &ix))) die(ctx, "pgr_index_load_mdb", rc);
        std::ifstream f(pos[0] + ".midx");
        if (!f) {
            fprintf(stderr, "pgr-query: can't open %s.midx\n", pos[0].c_str());
            return 1;
        }
        std::string line;
        while (std::getline(f, line)) {  // sid \t len \t name \t source   (seq_db.rs:798-805)
            std::vector<std::string> c;
            size_t b = 0;
            for (size_t e; (e = line.find('\t', b)) != std::string::npos; b = e + 1) c.push_back(line.substr(b, e - b));
            c.push_back(line.substr(b));
            if (c.size() < 4) continue;
            seq_info[(uint32_t)strtoul(c[0].c_str(), nullptr, 10)] = SeqInfo{c[2], c[3] == "-" ? "N/A" : c[3], strtoull(c[1].c_str(), nullptr, 10)};
        }
    }
    // ---- queries, one batch
    const std::vector<pgrhost::SeqRec> queries = pgrhost::read_fastx(pos[1]);
    std::vector<const uint8_t *> qp;
    std::vector<uint64_t> ql;
    for (const auto &q : queries) {
        qp.push_back((const uint8_t *)q.seq.data());
        ql.push_back(q.seq.size());
    }
    // chains of the queries [q_base, q_base + res.n_queries) -> their files (the reference's loop body, rs:167-409)
    auto emit = [&](const pgr_hps_result &res, size_t q_base) -> int {
    for (size_t ql_ = 0; ql_ < res.n_queries; ++ql_) {
        const size_t qi = q_base + ql_;
        // chains -> regions per target (rs:167-285): only chains with more than 2 hit pairs; the forward / reverse
        // counters are NOT reset between the chains of a target (:171-183)
        std::map<uint32_t, std::vector<Region>> regions;
        for (uint64_t t = res.q_off[ql_]; t < res.q_off[ql_ + 1]; ++t) {
            const uint32_t sid = res.t_sid[t];
            uint64_t f_count = 0, r_count = 0;
            std::vector<Region> rg;
            for (uint64_t c = res.t_off[t]; c < res.t_off[t + 1]; ++c) {
                const uint64_t b = res.c_off[c], e = res.c_off[c + 1];
                if (e - b <= 2) continue;
                Region r;
                for (uint64_t h = b; h < e; ++h) {
                    const pgr_hitpair &x = res.hps[h];
                    r.aln.push_back(HP{x.qb, x.qe, x.qo, x.tb, x.te, x.to});
                    if (x.qo == x.to) ++f_count;
                    else ++r_count;
                }
                r.orientation = f_count > r_count ? 0u : 1u;
                std::vector<std::pair<uint32_t, uint32_t>> tc;
                for (const HP &h : r.aln) tc.emplace_back(h.tb, h.te);
                std::sort(tc.begin(), tc.end());
                r.bgn = tc.front().first;
                r.end = tc.back().second;
                r.len = r.end - r.bgn;
                rg.push_back(std::move(r));
            }
            if (rg.empty()) continue;
            std::vector<Region> merged;
            for (uint32_t ori = 0; ori < 2; ++ori) {  // merge per orientation when closer than merge_range_tol
                std::vector<Region> v;
                for (const Region &r : rg)
                    if (r.orientation == ori) v.push_back(r);
                std::sort(v.begin(), v.end());
                bool have = false;
                Region last;
                for (Region &r : v) {
                    if (!have) {
                        last = std::move(r);
                        have = true;
                    } else if ((int64_t)r.bgn - (int64_t)last.end < (int64_t)merge_tol) {
                        last.end = std::max(r.end, last.end);
                        last.len = last.end - last.bgn;
                        last.aln.insert(last.aln.end(), r.aln.begin(), r.aln.end());
                    } else {
                        merged.push_back(std::move(last));
                        last = std::move(r);
                    }
                }
                if (have && last.len > 0) merged.push_back(std::move(last));
            }
            regions[sid] = std::move(merged);
        }
        char ext[32];
        snprintf(ext, sizeof ext, bed_summary ? "%03zu.hit.bed" : "%03zu.hit", qi);
        FILE *hit = fopen(with_extension(pos[2], ext).c_str(), "w");
        if (!hit) {
            fprintf(stderr, "pgr-query: can't write %s\n", with_extension(pos[2], ext).c_str());
            return 1;
        }
        if (bed_summary)
            fputs("#target\tbgn\tend\tquery\tcolor\torientation\tq_len\taln_anchor_count\tq_idx\tsrc\tctg_bgn\tctg_end\n", hit);
        else
            fputs("#idx\tq_ctg_name\tq_ctg_bgn\tq_ctg_end\tq_ctg_len\taln_anchor_count\tsrc\tctg\tctg_bgn\tctg_end\torientation\tctg_name\n", hit);
        struct Fa {
            uint32_t sid, b, e, ori;
            std::string name;
        };
        std::vector<Fa> fa;
        const std::string &q_name = queries[qi].name;
        const size_t q_len = queries[qi].seq.size();
        for (auto &kv : regions) {
            const SeqInfo &si = seq_info[kv.first];
            const std::string base = stem(si.src);
            for (Region &r : kv.second) {
                std::sort(r.aln.begin(), r.aln.end());
                const uint32_t q_bgn = r.aln.front().qb, q_end = r.aln.back().qe;
                // (a std::string: contig names of assemblies can be long, a fixed buffer would cut them silently)
                const std::string tname_s = base + "::" + si.name + "_" + std::to_string(r.bgn) + "_" + std::to_string(r.end) + "_" +
                                            std::to_string(r.orientation);
                const char *tname = tname_s.c_str();
                if (bed_summary)
                    fprintf(hit, "%s\t%u\t%u\t%s\t#AAAAAA\t%u\t%zu\t%zu\t%zu\t%s\t%u\t%u\t%s\n", si.name.c_str(), r.bgn, r.end,
                            q_name.c_str(), r.orientation, q_len, r.aln.size(), qi, si.src.c_str(), q_bgn, q_end, tname);
                else
                    fprintf(hit, "%03zu\t%s\t%u\t%u\t%zu\t%zu\t%s\t%s\t%u\t%u\t%u\t%s\n", qi, q_name.c_str(), q_bgn, q_end, q_len,
                            r.aln.size(), si.src.c_str(), si.name.c_str(), r.bgn, r.end, r.orientation, tname);
                fa.push_back(Fa{kv.first, r.bgn, r.end, r.orientation, tname_s});
            }
        }
        fclose(hit);
        if (fastx_file && !only_summary) {
            char fe[32];
            snprintf(fe, sizeof fe, "%03zu.fa", qi);
            FILE *f = fopen(with_extension(pos[2], fe).c_str(), "w");
            if (!f) return 1;
            for (const Fa &x : fa) {
                const std::string &s = db_seqs[x.sid].seq;
                const size_t b = std::min<size_t>(x.b, s.size()), e = std::min<size_t>(x.e, s.size());
                std::string t = s.substr(b, e > b ? e - b : 0);
                if (x.ori == 1) t = reverse_complement(t);
                fprintf(f, ">%s\n%s\n", x.name.c_str(), t.c_str());
            }
            fclose(f);
        }
    }
    return 0;
    };
    if (queries.size() <= query_batch) {  // one batch, one call
        pgr_hps_result res;
        if ((rc = pgr_query_hps_batch(ctx, ix, (uint32_t)queries.size(), qp.data(), ql.data(), gap_penalty, max_count, max_q,
                                      max_t, max_span, 0, 0, 0, &res)))
            die(ctx, "pgr_query_hps_batch", rc);
        if (emit(res, 0)) return 1;
        pgr_hps_result_free(&res);
    } else {
        // batches of query_batch queries, two in flight (the reference loops over its queries with rayon, rs:135-138): batch i + 1
        // is staged and its tiles run while batch i's chains are looked up, chained and sent home
        pgr_spec ispec;
        if ((rc = pgr_index_spec(ix, &ispec))) die(ctx, "pgr_index_spec", rc);
        pgr_pipe *pipe = nullptr;
        if ((rc = pgr_pipe_create(ctx, &ispec, &pipe))) die(ctx, "pgr_pipe_create", rc);
        std::vector<std::pair<pgr_batch *, size_t>> flying;  // (resident batch, first query), oldest first
        auto collect_one = [&]() -> int {
            pgr_hps_result res;
            if ((rc = pgr_pipe_collect_query(pipe, &res))) die(ctx, "pgr_pipe_collect_query", rc);
            const int bad = emit(res, flying.front().second);
            pgr_hps_result_free(&res);
            pgr_batch_destroy(flying.front().first);
            flying.erase(flying.begin());
            return bad;
        };
        for (size_t q0 = 0; q0 < queries.size(); q0 += query_batch) {
            const size_t nb = std::min(query_batch, queries.size() - q0);
            pgr_batch *qb = nullptr;
            if ((rc = pgr_batch_from_ascii(ctx, (uint32_t)nb, qp.data() + q0, ql.data() + q0, &qb))) die(ctx, "pgr_batch_from_ascii", rc);
            if (flying.size() == 2 && collect_one()) return 1;
            if ((rc = pgr_pipe_submit_query(pipe, qb, ix, gap_penalty, max_count, max_q, max_t, max_span, 0, 0, 0)))
                die(ctx, "pgr_pipe_submit_query", rc);
            flying.emplace_back(qb, q0);
        }
        while (!flying.empty())
            if (collect_one()) return 1;
        pgr_pipe_destroy(pipe);
    }
    pgr_index_destroy(ix);
    pgr_ctx_destroy(ctx);
    return 0;
}
