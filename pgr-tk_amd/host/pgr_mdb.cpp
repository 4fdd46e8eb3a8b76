// pgr-mdb counterpart (pgr-bin/src/bin/pgr-mdb.rs:26-111) in C++ above the C ABI of libpgrhip.so:
//   pgr-mdb <filelist> <prefix> [-w 80] [-k 56] [-r 4] [-m 64] [--sketch] [--batch-bp N] [--reference-sid-quirk]
//           [--ranks N [--devices 0,1,...]]
//   pgr-mdb --synthetic NxL --seed S <prefix> [the same options] [--write-fasta <path>]
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
#include <fcntl.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>

#include "fastx.hpp"
#include "pgr_hip.h"

// --ranks N: the sharded build of SURVEY.md section 8e without Python.  One process per GPU: the parent forks and every child
// EXECs this program again with --as-rank (a rank is a fresh process image: nothing of the parent's runtime state -- locks held by
// threads a loaded library may have started, half-initialised HIP/HSA singletons -- is inherited across the fork), dies with
// its parent (PR_SET_PDEATHSIG) and the parent hands SIGTERM/SIGINT on to its ranks; every rank reads the inputs, takes its contigs from the greedy length-balanced partition (the unit of
// parallelism is the contig, pgr-db/src/seq_db.rs:460-467), computes their pair records with global sequence ids and
// takes part in pgr_exchange_shard_records round after round: the frag_map (seq_db.rs:605-612) is key-range sharded,
// rank r owns the r-th range of first hashes and sorts only that.  Every rank writes its shard as <prefix>.mdb.rank<r>,
// the parent concatenates the shards in rank order into <prefix>.mdb (ascending ranges: the file is byte-identical to
// the single-process one), rank 0 writes the .midx.  The 128-byte RCCL unique id goes from rank 0 to the others through pipes.
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

// contigs -> pair records of `ix`.  Default: pgr_index_add_batch (the library packs the ASCII bytes on the host while it
// stages them).  --prepack: this program packs first (pgr_pack_ascii, all its CPUs) and hands over 2-bit planes
// (pgr_index_add_packed): what a host does that keeps its sequences packed.
static bool prepack = false;
static int add_contigs(pgr_ctx *ctx, pgr_index *ix, uint32_t n, const std::vector<const uint8_t *> &ptrs,
                       const std::vector<uint64_t> &lens, const std::vector<uint32_t> &sids, bool packed_first) {
    if (!packed_first) return pgr_index_add_batch(ctx, ix, n, ptrs.data(), lens.data(), sids.data());
    const uint64_t words = pgr_packed_words(n, lens.data());
    std::vector<uint64_t> planes((size_t)words);
    std::vector<uint32_t> valid((size_t)words);
    const int rc = pgr_pack_ascii(n, ptrs.data(), lens.data(), 0, planes.data(), valid.data(), nullptr);
    if (rc) return rc;
    return pgr_index_add_packed(ctx, ix, n, lens.data(), planes.data(), valid.data(), sids.data());
}

struct Synthetic {
    bool on = false;
    uint64_t n = 0, len = 0, seed = 0;
    std::string fasta_out;
    std::string name(uint64_t c) const { return "synth_" + std::to_string(seed) + "_" + std::to_string(c); }
    std::string source() const { return "synthetic:" + std::to_string(n) + "x" + std::to_string(len) + ":seed=" + std::to_string(seed); }
};
static Synthetic synth;

static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// the host form of the generator, only for --write-fasta
static bool write_synthetic_fasta(const Synthetic &sy) {
    FILE *f = fopen(sy.fasta_out.c_str(), "w");
    if (!f) return false;
    std::vector<char> line;
    bool ok = true;
    for (uint64_t c = 0; ok && c < sy.n; ++c) {
        ok = fprintf(f, ">%s\n", sy.name(c).c_str()) >= 0;
        line.resize((size_t)sy.len + 1);
        for (uint64_t i = 0; i < sy.len; i += 32) {
            const uint64_t z = splitmix64(sy.seed ^ (c * 0x9E3779B97F4A7C15ull) ^ (i >> 5));
            for (uint64_t j = 0; j < 32 && i + j < sy.len; ++j) line[(size_t)(i + j)] = "ACGT"[(z >> (2 * j)) & 3];
        }
        line[(size_t)sy.len] = '\n';
        ok = ok && fwrite(line.data(), 1, line.size(), f) == line.size();
    }
    return (fclose(f) == 0) && ok;
}

// synthetic contigs ids[0..n) -> pair records of `ix`, sequence id = contig id
static int add_synthetic(pgr_ctx *ctx, pgr_index *ix, const Synthetic &sy, const std::vector<uint64_t> &ids) {
    std::vector<uint64_t> lens(ids.size(), sy.len);
    std::vector<uint32_t> sids(ids.size());
    for (size_t i = 0; i < ids.size(); ++i) sids[i] = (uint32_t)ids[i];
    pgr_batch *b = nullptr;
    int rc = pgr_batch_synthetic_ids(ctx, (uint32_t)ids.size(), lens.data(), sy.seed, ids.data(), &b);
    if (rc) return rc;
    rc = pgr_index_add_resident(ctx, ix, b, sids.data());
    pgr_batch_destroy(b);
    return rc;
}

static void die(pgr_ctx *ctx, const char *what, int rc) {
    const char *msg = ctx ? pgr_last_error(ctx) : pgr_last_error(nullptr);
    fprintf(stderr, "pgr-mdb: %s failed (%d): %s\n", what, rc, msg);
    if (msg && strstr(msg, "did not") && strstr(msg, "within"))  // the library's watchdog: a peer that is slow is not a peer that is gone
        fprintf(stderr, "pgr-mdb: a rank waited longer than its time-out for the others.  If the ranks are merely unbalanced (one much "
                        "larger share of the contigs) raise PGR_EXCHANGE_COLLECTIVE_TIMEOUT_S (seconds; 0 = wait for ever) -- "
                        "PGR_EXCHANGE_TIMEOUT_S bounds the rendezvous of the ranks at start-up; a rank that has died says so in its "
                        "own last line above.\n");
    exit(1);
}

static int run_rank(const RankEnv &env, const pgr_spec &spec, uint64_t batch_bp, bool sid_quirk,
                    const std::vector<std::string> &pos);
static int merge_mdb_shards(const std::string &prefix, int ranks);

static std::vector<pid_t> g_kids;  // the rank processes of --ranks (parent only; fixed before the handler is installed)
static void forward_signal(int sig) {
    for (pid_t k : g_kids)
        if (k > 0) kill(k, sig);
}

int main(int argc, char **argv) {
    pgr_spec spec = {80, 56, 4, 64, 0};
    uint64_t batch_bp = 2000000000ull;
    bool sid_quirk = false;
    int ranks = 1;
    bool force_exchange = false;
    int as_rank = -1, id_read_fd = -1;
    std::vector<int> devices, id_write_fds;
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
        else if (a == "--as-rank") as_rank = atoi(val("--as-rank"));          // internal: this process IS rank r of --ranks N
        else if (a == "--id-read-fd") id_read_fd = atoi(val("--id-read-fd"));  // internal: the unique id arrives here
        else if (a == "--id-write-fds") {                                      // internal (rank 0): one pipe per other rank
            std::string v = val("--id-write-fds");
            for (size_t p = 0; p < v.size();) {
                const size_t q = v.find(',', p);
                id_write_fds.push_back(atoi(v.substr(p, q == std::string::npos ? std::string::npos : q - p).c_str()));
                if (q == std::string::npos) break;
                p = q + 1;
            }
        }
        else if (a == "--force-exchange") force_exchange = true;
        else if (a == "--prepack") prepack = true;
        else if (a == "--synthetic") {
            const std::string v = val("--synthetic");
            const size_t x = v.find_first_of("xX");
            char *e0 = nullptr, *e1 = nullptr;
            synth.n = strtoull(v.c_str(), &e0, 10);
            synth.len = x == std::string::npos ? 0 : strtoull(v.c_str() + x + 1, &e1, 10);
            if (x == std::string::npos || e0 != v.c_str() + x || !e1 || *e1 || !synth.n || !synth.len || synth.n > 0xFFFFFFFFull) {
                fprintf(stderr, "pgr-mdb: --synthetic wants NxL (contigs x bases per contig), e.g. 10x1000000\n");
                return 2;
            }
            synth.on = true;
        } else if (a == "--seed") synth.seed = strtoull(val("--seed"), nullptr, 10);
        else if (a == "--write-fasta") synth.fasta_out = val("--write-fasta");
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
    if (synth.on && pos.size() == 1) pos.insert(pos.begin(), std::string());  // no <filelist> with --synthetic
    if (pos.size() != 2 || ranks < 1 || (synth.on && !pos[0].empty())) {
        fprintf(stderr, "usage: pgr-mdb <filelist> <prefix> [-w 80 -k 56 -r 4 -m 64 --sketch] [--prepack] [--ranks N [--devices 0,1,..]]\n"
                        "       pgr-mdb --synthetic NxL --seed S <prefix> [...] [--write-fasta <path>]\n");
        return 2;
    }
    if (as_rank < 0 && synth.on && !synth.fasta_out.empty() && !write_synthetic_fasta(synth)) {
        fprintf(stderr, "pgr-mdb: can't write %s\n", synth.fasta_out.c_str());
        return 1;
    }
    if (sid_quirk && (ranks > 1 || force_exchange)) {
        // per-input sid restarts make sids ambiguous across files: the sharded build tells contigs apart by their sid, two
        // contigs of different files with the same sid next to each other in a rank's list would be fused into one
        fprintf(stderr, "pgr-mdb: --reference-sid-quirk cannot be combined with --ranks / --force-exchange\n");
        return 2;
    }
    if (as_rank >= 0) {  // a rank process (see the fork/exec below)
        if (as_rank >= ranks) {
            fprintf(stderr, "pgr-mdb: --as-rank %d of %d ranks\n", as_rank, ranks);
            return 2;
        }
        RankEnv env;
        env.rank = as_rank;
        env.world = ranks;
        env.force_exchange = force_exchange;
        env.device = devices.empty() ? as_rank : devices[(size_t)as_rank % devices.size()];
        env.id_read_fd = id_read_fd;
        env.id_write_fds = id_write_fds;
        return run_rank(env, spec, batch_bp, sid_quirk, pos);
    }
    if (ranks == 1 && devices.empty() && !force_exchange) {
        RankEnv env;
        return run_rank(env, spec, batch_bp, sid_quirk, pos);
    }
    // one process per GPU: fork, then exec this program again as --as-rank r (the pipe ends a rank needs stay open across the exec)
    std::vector<std::pair<int, int>> pipes((size_t)ranks, {-1, -1});
    for (int r = 1; r < ranks; ++r) {
        int fd[2];
        if (pipe(fd) != 0) {
            perror("pgr-mdb: pipe");
            return 1;
        }
        pipes[(size_t)r] = {fd[0], fd[1]};
    }
    const pid_t parent = getpid();
    std::vector<pid_t> kids;
    for (int r = 0; r < ranks; ++r) {
        const pid_t pid = fork();
        if (pid < 0) {
            perror("pgr-mdb: fork");
            for (pid_t o : kids) kill(o, SIGTERM);
            return 1;
        }
        if (pid == 0) {
            // a rank never outlives the program that started it (a killed parent must not leave ranks on the GPUs)
            (void)prctl(PR_SET_PDEATHSIG, SIGKILL);
            if (getppid() != parent) _exit(1);  // (the parent was gone before the prctl)
            std::vector<std::string> av(argv, argv + argc);
            av.push_back("--as-rank");
            av.push_back(std::to_string(r));
            std::string wfds;
            for (int q = 1; q < ranks; ++q) {
                if (r == 0) {
                    close(pipes[(size_t)q].first);
                    wfds += (wfds.empty() ? "" : ",") + std::to_string(pipes[(size_t)q].second);
                } else if (q == r) {
                    close(pipes[(size_t)q].second);
                    av.push_back("--id-read-fd");
                    av.push_back(std::to_string(pipes[(size_t)q].first));
                } else {
                    close(pipes[(size_t)q].first);
                    close(pipes[(size_t)q].second);
                }
            }
            if (!wfds.empty()) {
                av.push_back("--id-write-fds");
                av.push_back(wfds);
            }
            std::vector<char *> cav;
            for (auto &a : av) cav.push_back(const_cast<char *>(a.c_str()));
            cav.push_back(nullptr);
            execv("/proc/self/exe", cav.data());
            perror("pgr-mdb: exec of the rank process");
            _exit(127);
        }
        kids.push_back(pid);
    }
    g_kids = kids;
    {  // SIGTERM / SIGINT to the parent is handed on to the ranks (the wait loop below then sees them exit)
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_handler = forward_signal;
        sigaction(SIGTERM, &sa, nullptr);
        sigaction(SIGINT, &sa, nullptr);
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
    if (bad) return bad;
    return merge_mdb_shards(pos[1], ranks);
}

// <prefix>.mdb.rank0 .. rank<N-1> (each a complete .mdb of one key range, keys ascending) -> <prefix>.mdb
static int merge_mdb_shards(const std::string &prefix, int ranks) {
    const std::string final_path = prefix + ".mdb", tmp_path = final_path + ".tmp";
    std::vector<std::string> parts;
    uint64_t n_keys = 0;
    unsigned char hdr0[31];
    for (int r = 0; r < ranks; ++r) {
        parts.push_back(prefix + ".mdb.rank" + std::to_string(r));
        FILE *f = fopen(parts.back().c_str(), "rb");
        unsigned char hdr[31];
        if (!f || fread(hdr, 1, 31, f) != 31 || memcmp(hdr, "mdb", 3) != 0) {
            if (f) fclose(f);
            fprintf(stderr, "pgr-mdb: shard file %s missing or damaged\n", parts.back().c_str());
            return 1;
        }
        fclose(f);
        if (r == 0) memcpy(hdr0, hdr, 31);
        else if (memcmp(hdr0, hdr, 23) != 0) {
            fprintf(stderr, "pgr-mdb: shard files disagree on the ShmmrSpec\n");
            return 1;
        }
        uint64_t nk;
        memcpy(&nk, hdr + 23, 8);
        n_keys += nk;
    }
    FILE *o = fopen(tmp_path.c_str(), "wb");
    bool ok = o != nullptr;
    if (ok) {
        memcpy(hdr0 + 23, &n_keys, 8);
        ok = fwrite(hdr0, 1, 31, o) == 31;
    }
    std::vector<char> buf(4u << 20);
    for (size_t r = 0; ok && r < parts.size(); ++r) {
        FILE *f = fopen(parts[r].c_str(), "rb");
        ok = f && fseek(f, 31, SEEK_SET) == 0;
        size_t got;
        while (ok && (got = fread(buf.data(), 1, buf.size(), f)) > 0) ok = fwrite(buf.data(), 1, got, o) == got;
        if (f) {
            ok = ok && !ferror(f);
            fclose(f);
        }
    }
    if (o) ok = (fclose(o) == 0) && ok;
    if (ok) ok = rename(tmp_path.c_str(), final_path.c_str()) == 0;
    for (const auto &pth : parts) (void)remove(pth.c_str());
    if (!ok) {
        (void)remove(tmp_path.c_str());
        fprintf(stderr, "pgr-mdb: can't write %s\n", final_path.c_str());
        return 1;
    }
    fprintf(stderr, "%d key-range shards, %llu keys -> %s\n", ranks, (unsigned long long)n_keys, final_path.c_str());
    return 0;
}

static int run_rank(const RankEnv &env, const pgr_spec &spec, uint64_t batch_bp, bool sid_quirk,
                    const std::vector<std::string> &pos) {
    pgr_ctx *ctx = nullptr;
    int rc = pgr_ctx_create(env.device, &ctx);
    if (rc) die(nullptr, "pgr_ctx_create", rc);
    const bool owner = env.rank == 0;  // writes the .midx (and, without an exchange, the .mdb)
    pgr_index *ix = nullptr;  // plain build: the frag_map; sharded build: this rank's key range of it
    if ((rc = pgr_index_create(ctx, &spec, &ix))) die(ctx, "pgr_index_create", rc);
    pgr_exchange *xch = nullptr;
    if (env.world > 1 || env.force_exchange) {
        uint8_t id[PGR_UNIQUE_ID_BYTES] = {0};
        int64_t rccl_world1 = 0;
        (void)pgr_ctx_get_option(ctx, "exchange_rccl_world1", &rccl_world1);
        if (env.rank == 0 && (env.world > 1 || rccl_world1)) {  // (an exchange of one rank runs without RCCL and needs no id)
            if ((rc = pgr_exchange_unique_id(ctx, id))) die(ctx, "pgr_exchange_unique_id", rc);
            for (int fd : env.id_write_fds)
                if (write(fd, id, sizeof(id)) != (ssize_t)sizeof(id)) {
                    perror("pgr-mdb: write unique id");
                    return 1;
                }
        } else if (env.rank > 0) {
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

    std::ifstream fl;
    if (!synth.on) {
        fl.open(pos[0]);
        if (!fl) {
            fprintf(stderr, "pgr-mdb: can't open %s\n", pos[0].c_str());
            return 1;
        }
    }
    struct Midx {
        uint32_t sid;
        size_t len;
        std::string name, src;
    };
    std::vector<Midx> midx;
    uint32_t sid = 0;
    std::string path;
    const uint64_t per_batch = synth.on ? std::max<uint64_t>(1, batch_bp / synth.len) : 0;  // synthetic contigs per GPU batch
    if (synth.on)
        for (uint64_t c = 0; c < synth.n; ++c) midx.push_back(Midx{(uint32_t)c, (size_t)synth.len, synth.name(c), synth.source()});
    if (synth.on && !xch) {
        // the loader's loop as a software pipeline: two batches in flight, the list stage and the pair
        // records of batch i beside the tiles of batch i + 1; the append buffer sized from the spec's density up front
        if (spec.w == 80 && spec.k == 56 && spec.r == 4 && !spec.sketch)
            (void)pgr_index_reserve(ctx, ix, (uint64_t)((double)synth.n * (double)synth.len * 0.00304 * 1.02) + 4096);
        pgr_pipe *pipe = nullptr;
        if ((rc = pgr_pipe_create(ctx, &spec, &pipe))) die(ctx, "pgr_pipe_create", rc);
        std::vector<pgr_batch *> alive;  // a batch outlives its job
        auto collect_one = [&]() {
            if ((rc = pgr_pipe_collect(pipe, nullptr, nullptr))) die(ctx, "pgr_pipe_collect", rc);
            pgr_batch_destroy(alive.front());
            alive.erase(alive.begin());
        };
        for (uint64_t c = 0; c < synth.n; c += per_batch) {
            std::vector<uint64_t> ids, lens;
            std::vector<uint32_t> sids;
            for (uint64_t q = c; q < std::min(synth.n, c + per_batch); ++q) {
                ids.push_back(q);
                lens.push_back(synth.len);
                sids.push_back((uint32_t)q);
            }
            pgr_batch *b = nullptr;
            if ((rc = pgr_batch_synthetic_ids(ctx, (uint32_t)ids.size(), lens.data(), synth.seed, ids.data(), &b))) die(ctx, "pgr_batch_synthetic_ids", rc);
            if (pgr_pipe_in_flight(pipe) == 2) collect_one();
            if ((rc = pgr_pipe_submit(pipe, b, sids.data(), ix, nullptr, 0))) die(ctx, "pgr_pipe_submit", rc);
            alive.push_back(b);
        }
        while (pgr_pipe_in_flight(pipe) > 0) collect_one();
        pgr_pipe_destroy(pipe);
    } else if (!xch) {
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
                if ((rc = add_contigs(ctx, ix, (uint32_t)(j - i), ptrs, lens, sids, prepack))) die(ctx, "pgr_index_add_batch", rc);
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
        if (synth.on)
            for (uint64_t c = 0; c < synth.n; ++c) {
                lens_all.push_back(synth.len);
                sids_all.push_back((uint32_t)c);
            }
        while (!synth.on && std::getline(fl, path)) {
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
        uint64_t sent_total = 0, received_total = 0;
        for (size_t round = 0; round < n_rounds; ++round) {
            // this round's pair records of this rank (an index object is the library's container for device records)
            pgr_index *part = nullptr;
            if ((rc = pgr_index_create(ctx, &spec, &part))) die(ctx, "pgr_index_create", rc);
            if (i < mine.size()) {
                size_t j = i;
                uint64_t tot = 0;
                while (j < mine.size() && (j == i || tot + lens_all[mine[j]] <= batch_bp)) tot += lens_all[mine[j++]];
                if (synth.on) {
                    std::vector<uint64_t> ids;
                    for (size_t q = i; q < j; ++q) ids.push_back((uint64_t)mine[q]);
                    if ((rc = add_synthetic(ctx, part, synth, ids))) die(ctx, "pgr_batch_synthetic_ids / pgr_index_add_resident", rc);
                } else {
                    std::vector<const uint8_t *> ptrs;
                    std::vector<uint64_t> lens;
                    std::vector<uint32_t> sids;
                    for (size_t q = i; q < j; ++q) {
                        ptrs.push_back((const uint8_t *)all[mine[q]].seq.data());
                        lens.push_back(lens_all[mine[q]]);
                        sids.push_back(sids_all[mine[q]]);
                    }
                    if ((rc = add_contigs(ctx, part, (uint32_t)(j - i), ptrs, lens, sids, prepack))) die(ctx, "pgr_index_add_batch", rc);
                }
                i = j;
            }
            const uint64_t n_part = pgr_index_n_records(part);
            uint64_t got = 0;
            // the key ranges are fixed by the first round's pooled sample and kept for the later rounds
            if ((rc = pgr_exchange_shard_records(xch, pgr_index_device_records(part), n_part, ix, round > 0, nullptr, &got)))
                die(ctx, "pgr_exchange_shard_records", rc);
            sent_total += n_part;
            received_total += got;
            pgr_index_destroy(part);
        }
        fprintf(stderr, "rank %d/%d (device %d): %zu of %zu contigs, %zu exchange rounds, %llu pair records sent, %llu in its key range\n",
                env.rank, env.world, env.device, mine.size(), lens_all.size(), n_rounds, (unsigned long long)sent_total,
                (unsigned long long)received_total);
    }
    const bool sharded = xch != nullptr;
    if (xch) pgr_exchange_destroy(xch);
    if ((rc = pgr_index_finalize(ctx, ix))) die(ctx, "pgr_index_finalize", rc);
    const std::string mdb_path = sharded ? pos[1] + ".mdb.rank" + std::to_string(env.rank) : pos[1] + ".mdb";
    if ((rc = pgr_index_write_mdb(ctx, ix, mdb_path.c_str()))) die(ctx, "pgr_index_write_mdb", rc);
    if (!owner) {
        pgr_index_destroy(ix);
        pgr_ctx_destroy(ctx);
        return 0;
    }
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
