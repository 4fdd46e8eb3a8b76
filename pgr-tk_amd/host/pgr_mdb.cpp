// pgr-mdb counterpart (pgr-bin/src/bin/pgr-mdb.rs:26-111) in C++ above the C ABI of libpgrhip.so:
//   pgr-mdb <filelist> <prefix> [-w 80] [-k 56] [-r 4] [-m 64] [--sketch] [--batch-bp N] [--reference-sid-quirk]
//           [--ranks N [--devices 0,1,...]]
// builds <prefix>.mdb + <prefix>.midx.  The reference iterates an AGC archive; AGC is not available here, so
// <filelist> lists FASTA / FASTQ (.gz) files.  Index-only path (seq_db.rs:541-615): fragment id = pair ordinal in
// the contig; the host owns sequence iteration and the .midx, the GPU computes shimmers and the frag_map.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include <cerrno>
#include <csignal>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>

#include "fastx.hpp"
#include "pgr_hip.h"

// --ranks N: the sharded build of SURVEY.md section 8e without Python.  One process per GPU (forked BEFORE the first HIP
// call); every rank reads the inputs, takes its contigs from the greedy length-balanced partition (the unit of
// parallelism is the contig, pgr-db/src/seq_db.rs:460-467), computes their shimmers with global sequence ids and takes
// part in pgr_exchange_gather_into_index round after round; rank 0 owns the frag_map (seq_db.rs:605-612) and writes
// the files.  The 128-byte RCCL unique id goes from rank 0 to the others through pipes.
struct RankEnv {
    int rank = 0, world = 1, device = 0;
    bool force_exchange = false;        // run the exchange code path even with one rank (plumbing test on a 1-GPU box)
    int id_read_fd = -1;                // ranks > 0: read the unique id here
    std::vector<int> id_write_fds;      // rank 0: write it to every other rank
};

static std::vector<std::vector<size_t>> shard_by_length(const std::vector<uint64_t> &lens, int world) {
    std::vector<size_t> order(lens.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return lens[a] > lens[b]; });
    std::vector<uint64_t> load((size_t)world, 0);
    std::vector<std::vector<size_t>> shards((size_t)world);
    for (size_t i : order) {
        size_t r = 0;
        for (size_t q = 1; q < (size_t)world; ++q)
            if (load[q] < load[r]) r = q;
        shards[r].push_back(i);
        load[r] += lens[i];
    }
    for (auto &sh : shards) std::sort(sh.begin(), sh.end());  // file order inside a rank
    return shards;
}

static void die(pgr_ctx *ctx, const char *what, int rc) {
    fprintf(stderr, "pgr-mdb: %s failed (%d): %s\n", what, rc, ctx ? pgr_last_error(ctx) : pgr_last_error(nullptr));
    exit(1);
}

static int run_rank(const RankEnv &env, const pgr_spec &spec, uint64_t batch_bp, bool sid_quirk,
                    const std::vector<std::string> &pos);

int main(int argc, char **argv) {
    pgr_spec spec = {80, 56, 4, 64, 0};
    uint64_t batch_bp = 2000000000ull;
    bool sid_quirk = false;
    int ranks = 1;
    bool force_exchange = false;
    std::vector<int> devices;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&](const char *name) -> const char * {
            if (i + 1 >= argc) {
                fprintf(stderr, "pgr-mdb: %s needs a value\n", name);
                exit(2);
            }
            return argv[++i];
        };
        if (a == "-w") spec.w = (uint32_t)atoi(val("-w"));
        else if (a == "-k") spec.k = (uint32_t)atoi(val("-k"));
        else if (a == "-r") spec.r = (uint32_t)atoi(val("-r"));
        else if (a == "-m" || a == "--min-span") spec.min_span = (uint32_t)atoi(val("-m"));
        else if (a == "--sketch") spec.sketch = 1;
        else if (a == "--batch-bp") batch_bp = strtoull(val("--batch-bp"), nullptr, 10);
        else if (a == "--reference-sid-quirk") sid_quirk = true;  // load_index_from_reader restarts at 0 per input (seq_db.rs:543)
        else if (a == "--ranks") ranks = atoi(val("--ranks"));
        else if (a == "--force-exchange") force_exchange = true;
        else if (a == "--devices") {
            std::string v = val("--devices");
            for (size_t p = 0; p <= v.size();) {
                const size_t q = v.find(',', p);
                devices.push_back(atoi(v.substr(p, q == std::string::npos ? std::string::npos : q - p).c_str()));
                if (q == std::string::npos) break;
                p = q + 1;
            }
        } else pos.push_back(a);
    }
    if (pos.size() != 2 || ranks < 1) {
        fprintf(stderr, "usage: pgr-mdb <filelist> <prefix> [-w 80 -k 56 -r 4 -m 64 --sketch] [--ranks N [--devices 0,1,..]]\n");
        return 2;
    }
    if (sid_quirk && (ranks > 1 || force_exchange)) {
        // per-input sid restarts make sids ambiguous across files: the sharded build tells contigs apart by their sid, two
        // contigs of different files with the same sid next to each other in a rank's list would be fused into one
        fprintf(stderr, "pgr-mdb: --reference-sid-quirk cannot be combined with --ranks / --force-exchange\n");
        return 2;
    }
    if (ranks == 1 && devices.empty() && !force_exchange) {
        RankEnv env;
        return run_rank(env, spec, batch_bp, sid_quirk, pos);
    }
    // one process per GPU, forked before anything touches HIP
    std::vector<std::pair<int, int>> pipes((size_t)ranks, {-1, -1});
    for (int r = 1; r < ranks; ++r) {
        int fd[2];
        if (pipe(fd) != 0) {
            perror("pgr-mdb: pipe");
            return 1;
        }
        pipes[(size_t)r] = {fd[0], fd[1]};
    }
    std::vector<pid_t> kids;
    for (int r = 0; r < ranks; ++r) {
        const pid_t pid = fork();
        if (pid < 0) {
            perror("pgr-mdb: fork");
            return 1;
        }
        if (pid == 0) {
            RankEnv env;
            env.rank = r;
            env.world = ranks;
            env.force_exchange = force_exchange;
            env.device = devices.empty() ? r : devices[(size_t)r % devices.size()];
            for (int q = 1; q < ranks; ++q) {
                if (r == 0) {
                    close(pipes[(size_t)q].first);
                    env.id_write_fds.push_back(pipes[(size_t)q].second);
                } else if (q == r) {
                    close(pipes[(size_t)q].second);
                    env.id_read_fd = pipes[(size_t)q].first;
                } else {
                    close(pipes[(size_t)q].first);
                    close(pipes[(size_t)q].second);
                }
            }
            _exit(run_rank(env, spec, batch_bp, sid_quirk, pos));
        }
        kids.push_back(pid);
    }
    for (int r = 1; r < ranks; ++r) {
        close(pipes[(size_t)r].first);
        close(pipes[(size_t)r].second);
    }
    // a rank that dies between two collectives leaves the others blocked in RCCL for ever: at the first abnormal exit the
    // remaining ranks are terminated and the build fails
    int bad = 0;
    size_t left = kids.size();
    while (left) {
        int st = 0;
        const pid_t k = waitpid(-1, &st, 0);
        if (k < 0) {
            if (errno == EINTR) continue;
            bad = 1;
            break;
        }
        auto it = std::find(kids.begin(), kids.end(), k);
        if (it == kids.end()) continue;
        *it = -1;
        --left;
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
            if (!bad) {
                fprintf(stderr, "pgr-mdb: a rank process failed; terminating the other ranks\n");
                for (pid_t o : kids)
                    if (o > 0) kill(o, SIGTERM);
            }
            bad = 1;
        }
    }
    return bad;
}

static int run_rank(const RankEnv &env, const pgr_spec &spec, uint64_t batch_bp, bool sid_quirk,
                    const std::vector<std::string> &pos) {
    pgr_ctx *ctx = nullptr;
    int rc = pgr_ctx_create(env.device, &ctx);
    if (rc) die(nullptr, "pgr_ctx_create", rc);
    const bool owner = env.rank == 0;  // owns the frag_map and writes the files
    pgr_index *ix = nullptr;
    if (owner && (rc = pgr_index_create(ctx, &spec, &ix))) die(ctx, "pgr_index_create", rc);
    pgr_exchange *xch = nullptr;
    if (env.world > 1 || env.force_exchange) {
        uint8_t id[PGR_UNIQUE_ID_BYTES];
        if (env.rank == 0) {
            if ((rc = pgr_exchange_unique_id(ctx, id))) die(ctx, "pgr_exchange_unique_id", rc);
            for (int fd : env.id_write_fds)
                if (write(fd, id, sizeof(id)) != (ssize_t)sizeof(id)) {
                    perror("pgr-mdb: write unique id");
                    return 1;
                }
        } else {
            size_t got = 0;
            while (got < sizeof(id)) {
                const ssize_t n = read(env.id_read_fd, id + got, sizeof(id) - got);
                if (n <= 0) {
                    fprintf(stderr, "pgr-mdb: rank %d got no unique id from rank 0\n", env.rank);
                    return 1;
                }
                got += (size_t)n;
            }
        }
        if ((rc = pgr_exchange_create(ctx, id, env.rank, env.world, &xch))) die(ctx, "pgr_exchange_create", rc);
    }

    std::ifstream fl(pos[0]);
    if (!fl) {
        fprintf(stderr, "pgr-mdb: can't open %s\n", pos[0].c_str());
        return 1;
    }
    struct Midx {
        uint32_t sid;
        size_t len;
        std::string name, src;
    };
    std::vector<Midx> midx;
    uint32_t sid = 0;
    std::string path;
    if (!xch) {
        while (std::getline(fl, path)) {
            while (!path.empty() && (path.back() == '\r' || path.back() == ' ')) path.pop_back();
            if (path.empty()) continue;
            const std::vector<pgrhost::SeqRec> recs = pgrhost::read_fastx(path);
            if (sid_quirk) sid = 0;
            size_t i = 0;
            while (i < recs.size()) {  // one GPU batch per ~batch_bp (the reference feeds <= 129 contigs, seq_db.rs:549-564)
                size_t j = i;
                uint64_t tot = 0;
                while (j < recs.size() && (j == i || tot + recs[j].seq.size() <= batch_bp)) tot += recs[j++].seq.size();
                std::vector<const uint8_t *> ptrs;
                std::vector<uint64_t> lens;
                std::vector<uint32_t> sids;
                for (size_t q = i; q < j; ++q) {
                    ptrs.push_back((const uint8_t *)recs[q].seq.data());
                    lens.push_back(recs[q].seq.size());
                    sids.push_back(sid + (uint32_t)(q - i));
                }
                if ((rc = pgr_index_add_batch(ctx, ix, (uint32_t)(j - i), ptrs.data(), lens.data(), sids.data())))
                    die(ctx, "pgr_index_add_batch", rc);
                for (size_t q = i; q < j; ++q) midx.push_back(Midx{sid++, recs[q].seq.size(), recs[q].name, path});
                i = j;
            }
        }
    } else {
        // sharded: every rank sees the whole contig list (sids in file order), works on its share, and all ranks run
        // the same number of exchange rounds
        std::vector<pgrhost::SeqRec> all;
        std::vector<uint64_t> lens_all;
        std::vector<uint32_t> sids_all;
        while (std::getline(fl, path)) {
            while (!path.empty() && (path.back() == '\r' || path.back() == ' ')) path.pop_back();
            if (path.empty()) continue;
            std::vector<pgrhost::SeqRec> recs = pgrhost::read_fastx(path);
            if (sid_quirk) sid = 0;
            for (auto &r : recs) {
                midx.push_back(Midx{sid, r.seq.size(), r.name, path});
                lens_all.push_back(r.seq.size());
                sids_all.push_back(sid++);
                all.push_back(std::move(r));
            }
        }
        const auto shards = shard_by_length(lens_all, env.world);
        auto rounds_of = [&](const std::vector<size_t> &sh) {  // batches of ~batch_bp, as the loop below cuts them
            size_t n = 0, i = 0;
            while (i < sh.size()) {
                uint64_t tot = 0;
                size_t j = i;
                while (j < sh.size() && (j == i || tot + lens_all[sh[j]] <= batch_bp)) tot += lens_all[sh[j++]];
                ++n;
                i = j;
            }
            return n;
        };
        size_t n_rounds = 0;
        for (const auto &sh : shards) n_rounds = std::max(n_rounds, rounds_of(sh));
        const std::vector<size_t> &mine = shards[(size_t)env.rank];
        size_t i = 0;
        uint64_t gathered_total = 0;
        for (size_t round = 0; round < n_rounds; ++round) {
            pgr_batch *b = nullptr;
            pgr_shmmrs *sh = nullptr;
            std::vector<uint32_t> rids;
            if (i < mine.size()) {
                size_t j = i;
                uint64_t tot = 0;
                while (j < mine.size() && (j == i || tot + lens_all[mine[j]] <= batch_bp)) tot += lens_all[mine[j++]];
                std::vector<const uint8_t *> ptrs;
                std::vector<uint64_t> lens;
                for (size_t q = i; q < j; ++q) {
                    ptrs.push_back((const uint8_t *)all[mine[q]].seq.data());
                    lens.push_back(lens_all[mine[q]]);
                    rids.push_back(sids_all[mine[q]]);
                }
                if ((rc = pgr_batch_from_ascii(ctx, (uint32_t)(j - i), ptrs.data(), lens.data(), &b))) die(ctx, "pgr_batch_from_ascii", rc);
                if ((rc = pgr_shmmrs_compute(ctx, b, &spec, nullptr, 0, &sh))) die(ctx, "pgr_shmmrs_compute", rc);
                i = j;
            }
            uint64_t n_g = 0;
            if ((rc = pgr_exchange_gather_into_index(xch, sh, rids.data(), ix, &n_g))) die(ctx, "pgr_exchange_gather_into_index", rc);
            gathered_total += n_g;
            pgr_shmmrs_destroy(sh);
            pgr_batch_destroy(b);
        }
        fprintf(stderr, "rank %d/%d (device %d): %zu of %zu contigs, %zu exchange rounds, %llu shimmers gathered\n", env.rank,
                env.world, env.device, mine.size(), all.size(), n_rounds, (unsigned long long)gathered_total);
    }
    if (xch) pgr_exchange_destroy(xch);
    if (!owner) {
        pgr_ctx_destroy(ctx);
        return 0;
    }
    if ((rc = pgr_index_finalize(ctx, ix))) die(ctx, "pgr_index_finalize", rc);
    if ((rc = pgr_index_write_mdb(ctx, ix, (pos[1] + ".mdb").c_str()))) die(ctx, "pgr_index_write_mdb", rc);
    {  // seq_db.rs:798-805; written to a temporary name and renamed, every write checked
        const std::string final_path = pos[1] + ".midx", tmp_path = final_path + ".tmp";
        FILE *f = fopen(tmp_path.c_str(), "w");
        bool ok = f != nullptr;
        for (size_t i = 0; ok && i < midx.size(); ++i) {
            const Midx &m = midx[i];
            ok = fprintf(f, "%u\t%zu\t%s\t%s\n", m.sid, m.len, m.name.c_str(), m.src.c_str()) >= 0;
        }
        if (f) ok = (fclose(f) == 0) && ok;
        if (ok) ok = rename(tmp_path.c_str(), final_path.c_str()) == 0;
        if (!ok) {
            (void)remove(tmp_path.c_str());
            fprintf(stderr, "pgr-mdb: can't write %s\n", final_path.c_str());
            return 1;
        }
    }
    fprintf(stderr, "%zu sequences, %llu shimmer pairs, %llu keys -> %s.mdb / .midx\n", midx.size(),
            (unsigned long long)pgr_index_n_records(ix), (unsigned long long)pgr_index_n_keys(ix), pos[1].c_str());
    pgr_index_destroy(ix);
    pgr_ctx_destroy(ctx);
    return 0;
}
