"""Command-line counterparts of the reference's callers of the hot path:

  python -m pgrtk_amd.cli mdb   <filelist> <prefix> [-w 80 -k 56 -r 4 -m 64 --sketch]
  python -m pgrtk_amd.cli mdb   --synthetic NxL --seed S <prefix> [...] [--write-fasta <path>]
        pgr-mdb (pgr-bin/src/bin/pgr-mdb.rs:26-111): builds <prefix>.mdb + <prefix>.midx.  The reference reads
        AGC archives; AGC is not available here, so <filelist> lists FASTA/FASTQ(.gz) files.  Index-only path
        (seq_db.rs:541-615): fragment id = pair ordinal in the contig.
  python -m pgrtk_amd.cli query <db> <query.fa> <out_prefix> [--fastx_file | (default) .mdb/.midx prefix] ...
        pgr-query (pgr-bin/src/bin/pgr-query.rs:17-409): chains -> per-target regions -> <out>.NNN.hit[.bed]
        (and <out>.NNN.fa with --fastx_file unless --only_summary).

  python -m pgrtk_amd.cli pbundle-decomp <fastx> <out_prefix> [-w 48 -k 56 -r 4 --min-span 12 --min-cov 0 ...]
        pgr-pbundle-decomp (pgr-bin/src/bin/pgr-pbundle-decomp.rs:139-531): MAP-graph principal bundles of the
        sequences of <fastx> and the bundle decomposition of every contig -> <out>.bed + <out>.ctg.summary.tsv.

Sequence iteration, range merging and file writing are host code, as in the reference; shimmers, the
frag_map and the chaining run on the GPU.
"""
import argparse
import os
import sys

import numpy as np

from .engine import Index, make_spec
from .seqindexdb import SeqIndexDB, read_fastx

_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTacgtNn", b"TGCAtgcaNn"):
    _COMP[_a] = _b


def with_extension(prefix, ext):
    """Rust Path::with_extension, which is how every reference CLI derives its output names from <output_prefix>
    (pgr-query.rs:291-302, pgr-pbundle-decomp.rs:294-359): an extension the prefix's file name already carries is
    replaced ("out.v1" -> "out.000.hit"); a leading dot alone is not an extension."""
    head, name = os.path.split(prefix)
    if name in ("", ".."):
        return prefix
    dot = name.rfind(".")
    if dot > 0:
        name = name[:dot]
    return os.path.join(head, name + "." + ext)


def reverse_complement(seq):
    """pgr-db/src/fasta_io.rs:26-44"""
    return _COMP[np.frombuffer(seq, dtype=np.uint8)][::-1].tobytes()


# ----------------------------------------------------------------------------- pgr-mdb
def synthetic_contig(seed, contig, length):
    """host form of the counter-based generator of BASELINE.md section 4 (the device form is pgr_batch_synthetic):
    base(c,i) = (splitmix64(seed ^ c*0x9E3779B97F4A7C15 ^ (i>>5)) >> (2*(i&31))) & 3 -> ASCII bytes"""
    m = (1 << 64) - 1
    words = np.arange((length + 31) // 32, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = words ^ np.uint64(seed ^ ((contig * 0x9E3779B97F4A7C15) & m))
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    codes = ((z[:, None] >> (np.uint64(2) * np.arange(32, dtype=np.uint64))[None, :]) & np.uint64(3)).astype(np.uint8)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[codes.reshape(-1)[:length]].tobytes()


def parse_synthetic(text):
    """'NxL' -> (N, L)"""
    n, _, ln = text.lower().partition("x")
    if not (n.isdigit() and ln.isdigit() and int(n) > 0 and int(ln) > 0):
        raise SystemExit("--synthetic wants NxL (contigs x bases per contig), e.g. 10x1000000")
    return int(n), int(ln)


def cmd_mdb_synthetic(args, spec):
    """pgr-mdb on N synthetic contigs of L bases generated on the device (BASELINE.json
    configs[0] is `--synthetic 10x1000000 --seed 1`); contig c is sequence c, named synth_<seed>_<c>"""
    from .engine import Batch
    n, ln = parse_synthetic(args.synthetic)
    if args.write_fasta:
        with open(args.write_fasta, "wb") as f:
            for c in range(n):
                f.write(b">synth_%d_%d\n" % (args.seed, c) + synthetic_contig(args.seed, c, ln) + b"\n")
    ix = Index(spec)
    per = max(1, args.batch_bp // ln)
    for c in range(0, n, per):
        ids = list(range(c, min(n, c + per)))
        b = Batch.synthetic([ln] * len(ids), args.seed, ctx=ix.ctx, contig_ids=ids)
        ix.add_resident(b, sids=ids)
        b.close()
    src = "synthetic:%dx%d:seed=%d" % (n, ln, args.seed)
    return ix, [(c, ln, "synth_%d_%d" % (args.seed, c), src) for c in range(n)]


def cmd_mdb(args):
    spec = make_spec(args.w, args.k, args.r, args.min_span, args.sketch)
    if args.synthetic:
        if args.prefix is not None:
            raise SystemExit("--synthetic takes <prefix> only (no <filelist>)")
        args.prefix = args.filepath
        ix, midx = cmd_mdb_synthetic(args, spec)
        paths = []
    else:
        if args.prefix is None:
            raise SystemExit("usage: mdb <filelist> <prefix>")
        ix = Index(spec)
        paths = [l.strip() for l in open(args.filepath) if l.strip()]
        midx = []
    sid = 0
    for path in paths:
        recs = read_fastx(path)
        if args.reference_sid_quirk:
            sid = 0  # load_index_from_reader restarts at 0 for every input (seq_db.rs:543)
        # the reference feeds batches of <= 129 contigs (seq_db.rs:549-564); one GPU batch per ~2 Gbp here
        i = 0
        while i < len(recs):
            j, tot = i, 0
            while j < len(recs) and (j == i or tot + len(recs[j][1]) <= args.batch_bp):
                tot += len(recs[j][1])
                j += 1
            ix.add_seqs([s for _, s in recs[i:j]], sids=list(range(sid, sid + (j - i))))
            for name, s in recs[i:j]:
                midx.append((sid, len(s), name, path))
                sid += 1
            i = j
    ix.finalize()
    ix.ctx.check(__import__("pgrtk_amd")._ffi.lib().pgr_index_write_mdb(ix.ctx.handle, ix._h, (args.prefix + ".mdb").encode()))
    with open(args.prefix + ".midx", "w") as f:  # seq_db.rs:798-805
        for sid_, ln, name, src in midx:
            f.write("%d\t%d\t%s\t%s\n" % (sid_, ln, name, src))
    print("%d sequences, %d shimmer pairs, %d keys -> %s.mdb / .midx" % (len(midx), ix.n_records, ix.n_keys, args.prefix),
          file=sys.stderr)


# ----------------------------------------------------------------------------- pgr-query post-processing
def chains_to_regions(query_result, merge_range_tol):
    """pgr-query.rs:167-285 for ONE query: [(sid, [(score, [hit pairs])])] -> {sid: [(bgn, end, len, orientation, aln)]}.
    Keeps the reference's quirks: only chains with more than 2 hit pairs; the forward / reverse counters are
    NOT reset between the chains of a target (:171-183); regions are merged per orientation when the gap to
    the previous region is below merge_range_tol."""
    sid_to_alns = {}
    for sid, alns in query_result:
        f_count = r_count = 0
        for _score, aln in alns:
            if len(aln) > 2:
                for hp in aln:
                    if hp[0][2] == hp[1][2]:
                        f_count += 1
                    else:
                        r_count += 1
                orientation = 0 if f_count > r_count else 1
                sid_to_alns.setdefault(sid, []).append((aln, orientation))
    out = {}
    for sid, alns in sid_to_alns.items():
        rgns = []
        for aln, orientation in alns:
            tc = sorted((hp[1][0], hp[1][1]) for hp in aln)
            bgn, end = tc[0][0], tc[-1][1]
            rgns.append((bgn, end, end - bgn, orientation, list(aln)))
        merged = []
        for ori in (0, 1):
            last = None
            for r in sorted((x for x in rgns if x[3] == ori), key=lambda x: (x[0], x[1], x[2], x[3], x[4])):
                if last is None:
                    last = r
                elif r[0] - last[1] < merge_range_tol:
                    end = max(r[1], last[1])
                    last = (last[0], end, end - last[0], last[3], last[4] + r[4])
                else:
                    merged.append(last)
                    last = r
            if last is not None and last[2] > 0:
                merged.append(last)
        out[sid] = merged
    return out


def cmd_query(args):
    if getattr(args, "frg_file", False):
        sys.stderr.write("pgr-query: --frg-file (the reference's .frg sequence store, pgr-query.rs:26-27) is not supported: use the "
                         ".mdb/.midx pair (default) or --fastx-file\n")
        return 2
    sdb = SeqIndexDB()
    seqs_by_sid = None
    if args.fastx_file:
        sdb.load_from_fastx(args.pgr_db_prefix, args.w, args.k, args.r, args.min_span)
        if not args.only_summary:
            seqs_by_sid = {i: s for i, (_, s) in enumerate(read_fastx(args.pgr_db_prefix))}
    else:
        sdb.load_from_mdb_index(args.pgr_db_prefix)
    queries = read_fastx(args.query_fastx_path)
    results = sdb.query_fragments_to_hps([s for _, s in queries], args.gap_penalty_factor, args.max_count,
                                         args.max_query_count, args.max_target_count, args.max_aln_chain_span, None, False)
    for idx, ((q_name, q_seq), qr) in enumerate(zip(queries, results)):
        regions = chains_to_regions(qr, args.merge_range_tol)
        ext = "%03d.hit.bed" % idx if args.bed_summary else "%03d.hit" % idx
        fa_recs = []
        with open(with_extension(args.output_prefix, ext), "w") as hit:
            if args.bed_summary:
                hit.write("#" + "\t".join(["target", "bgn", "end", "query", "color", "orientation", "q_len",
                                           "aln_anchor_count", "q_idx", "src", "ctg_bgn", "ctg_end"]) + "\n")
            else:
                hit.write("#" + "\t".join(["idx", "q_ctg_name", "q_ctg_bgn", "q_ctg_end", "q_ctg_len", "aln_anchor_count",
                                           "src", "ctg", "ctg_bgn", "ctg_end", "orientation", "ctg_name"]) + "\n")
            for sid in sorted(regions):
                ctg, src, _ = sdb.seq_info[sid]
                src = src if src is not None else "N/A"
                base = os.path.splitext(os.path.basename(src))[0]
                for b, e, _, orientation, aln in regions[sid]:
                    aln = sorted(aln)
                    q_bgn, q_end = aln[0][0][0], aln[-1][0][1]
                    tname = "%s::%s_%d_%d_%d" % (base, ctg, b, e, orientation)
                    if args.bed_summary:
                        hit.write("\t".join(str(v) for v in [ctg, b, e, q_name, "#AAAAAA", orientation, len(q_seq), len(aln),
                                                             idx, src, q_bgn, q_end, tname]) + "\n")
                    else:
                        hit.write("\t".join(str(v) for v in ["%03d" % idx, q_name, q_bgn, q_end, len(q_seq), len(aln), src,
                                                             ctg, b, e, orientation, tname]) + "\n")
                    fa_recs.append((sid, b, e, orientation, tname))
        if seqs_by_sid is not None:
            with open(with_extension(args.output_prefix, "%03d.fa" % idx), "w") as fa:
                for sid, b, e, orientation, tname in fa_recs:
                    t = seqs_by_sid[sid][b:e]
                    if orientation == 1:
                        t = reverse_complement(t)
                    fa.write(">%s\n%s\n" % (tname, t.decode("ascii", "replace")))


# ----------------------------------------------------------------------------- pgr-pbundle-decomp
def group_smps_by_principle_bundle_id(smps, bundle_length_cutoff, bundle_merge_distance):
    """pgr-pbundle-decomp.rs:61-137: runs of shimmer pairs on the same (bundle, direction) longer than
    bundle_length_cutoff, neighbouring runs of the same bundle merged when closer than bundle_merge_distance.
    smps = [((h0,h1,p0,p1,o), (bundle id, direction, position) | None)] -> [[(smp, bid, direction, bpos)]]"""
    parts, cur = [], []
    pre = None
    for smp, info in smps:
        if info is None:
            continue
        key = (info[0], 0 if smp[4] == info[1] else 1)
        if pre is not None and key != pre:
            if cur[-1][0][3] - cur[0][0][2] > bundle_length_cutoff:
                parts.append(cur)
            cur = []
        pre = key
        cur.append((smp, key[0], key[1], info[2]))
    if cur and cur[-1][0][3] - cur[0][0][2] > bundle_length_cutoff:
        parts.append(cur)
    merged = []
    for p in parts:
        if merged and merged[-1][-1][1] == p[0][1] and merged[-1][-1][2] == p[0][2] and \
                abs(p[0][0][2] - merged[-1][-1][0][3]) < bundle_merge_distance:
            merged[-1] = merged[-1] + p
        else:
            merged.append(p)
    return merged


def pbundle_partitions(seq_names, decomposition, bundle_length_cutoff, bundle_merge_distance):
    """per contig in name order (rs:343): [(sid, name, partitions, {bid: count})]"""
    by_sid = dict(decomposition)
    out = []
    for sid, name in sorted(seq_names.items(), key=lambda t: t[1]):
        parts = group_smps_by_principle_bundle_id(by_sid.get(sid, []), bundle_length_cutoff, bundle_merge_distance)
        cnt = {}
        for p in parts:
            cnt[p[0][1]] = cnt.get(p[0][1], 0) + 1
        out.append((sid, name, parts, cnt))
    return out


def pbundle_bed_lines(seq_names, decomposition, bundles_with_id, k, bundle_length_cutoff=2500, bundle_merge_distance=10000):
    """the body of <out>.bed (rs:355-394): ctg, bgn, end, bid:bundle size:direction:first pos:last pos:R|U"""
    size = {b[0]: len(b[2]) for b in bundles_with_id}
    lines = []
    for _sid, name, parts, cnt in pbundle_partitions(seq_names, decomposition, bundle_length_cutoff, bundle_merge_distance):
        for p in parts:
            bid = p[0][1]
            lines.append("%s\t%d\t%d\t%d:%d:%d:%d:%d:%s" % (name, p[0][0][2], p[-1][0][3] + k, bid, size[bid], p[0][2], p[0][3],
                                                            p[-1][3], "R" if cnt[bid] > 1 else "U"))
    return lines


def _f32(x):
    """Rust `{}` of an f32: shortest decimal that round-trips, no exponent, no trailing .0"""
    x = np.float32(x)
    if np.isnan(x):
        return "NaN"
    if np.isinf(x):
        return "inf" if x > 0 else "-inf"
    return np.format_float_positional(x, unique=True, trim="-")


def pbundle_summary_lines(seq_info, partitions, k):
    """<out>.ctg.summary.tsv (rs:396-530); seq_info = {sid: (name, source, len)}"""
    hdr = ["ctg", "length", "repeat_bundle_count", "repeat_bundle_sum", "repeat_bundle_percentage", "repeat_bundle_mean",
           "repeat_bundle_min", "repeat_bundle_max", "non_repeat_bundle_count", "non_repeat_bundle_sum",
           "non_repeat_bundle_percentage", "non_repeat_bundle_mean", "non_repeat_bundle_min", "non_repeat_bundle_max",
           "total_bundle_count", "total_bundle_coverage_percentage"]
    lines = ["#" + "\t".join(hdr)]
    f32 = np.float32
    for sid, name, parts, cnt in partitions:
        ln = seq_info[sid][2]
        rep, non = [], []
        for p in parts:
            # e - b - k with e = last end + k (rs:362-372)
            (rep if cnt[p[0][1]] > 1 else non).append(p[-1][0][3] - p[0][0][2])
        rs, ns = sum(rep) & 0xFFFFFFFF, sum(non) & 0xFFFFFFFF

        def stats(v, total):
            if not v:
                return ["NA", "NA", "NA"]
            return [_f32(f32(total) / f32(len(v))), str(min(v)), str(max(v))]
        rmean, rmin, rmax = stats(rep, rs)
        nmean, nmin, nmax = stats(non, ns)
        lines.append("\t".join([name, str(ln), str(len(rep)), str(rs), _f32(f32(100.0) * f32(rs) / f32(ln)), rmean, rmin, rmax,
                                str(len(non)), str(ns), _f32(f32(100.0) * f32(ns) / f32(ln)), nmean, nmin, nmax,
                                str(len(rep) + len(non)), _f32(f32(100.0) * f32((rs + ns) & 0xFFFFFFFF) / f32(ln))]))
    return lines


def cmd_pbundle_decomp(args):
    from . import pdb
    sdb = SeqIndexDB()
    if args.precomputed_bundles:
        # rs:155-218: the bundles and the vertex map come from a .pdb written by an earlier run; its parameters replace
        # the command line's (rs:239-244); nothing is computed from <fastx_path> but the decomposition of its sequences
        (args.w, args.k, args.r, args.min_span, args.min_branch_size, args.min_cov, bundles, vmap) = \
            pdb.read_pdb(args.precomputed_bundles)
        sdb._reset(args.w, args.k, args.r, args.min_span)
        dec, seq_info = None, None
    else:
        sdb.load_from_fastx(args.fastx_path, args.w, args.k, args.r, args.min_span)
        bundles, dec = sdb.get_principal_bundle_decomposition(args.min_cov, args.min_branch_size)
        seq_info = sdb.seq_info
        # every bundle vertex is a shimmer pair of <fastx_path>'s own sequences, so the vertex map
        # (get_vertex_map_from_principal_bundles, ext.rs:512-531) is exactly the annotation of their pairs
        vmap = {(smp[0], smp[1]): info for _, smps in dec for smp, info in smps if info is not None}
        pdb.write_pdb(with_extension(args.output_prefix, "pdb"), args.w, args.k, args.r, args.min_span, args.min_branch_size, args.min_cov,
                      bundles, vmap)  # rs:357-383
    if args.decomp_fastx_path or args.include or dec is None:
        # decomposition of other sequences with that vertex map (rs:247-292 + ext.rs:976-1014)
        recs = read_fastx(args.decomp_fastx_path or args.fastx_path)
        if args.include:
            want = set(l.strip() for l in open(args.include) if l.strip())
            recs = [r for r in recs if r[0] in want]
        seq_info = {i: (name, args.decomp_fastx_path or args.fastx_path, len(s)) for i, (name, s) in enumerate(recs)}
        from .engine import frag_recs_batch
        q = frag_recs_batch([s for _, s in recs], sdb._spec, query_side=True, ctx=sdb.ctx)
        dec = []
        for i, rr in enumerate(q):
            smps = [(int(r["h0"]), int(r["h1"]), int(r["bgn"]), int(r["end"]), int(r["orient"])) for r in rr]
            dec.append((i, [(v, vmap.get((v[0], v[1]))) for v in smps]))
    names = {sid: v[0] for sid, v in seq_info.items()}
    parts = pbundle_partitions(names, dec, args.bundle_length_cutoff, args.bundle_merge_distance)
    with open(with_extension(args.output_prefix, "bed"), "w") as f:
        f.write("# cmd: %s\n" % " ".join(sys.argv))
        for line in pbundle_bed_lines(names, dec, bundles, args.k, args.bundle_length_cutoff, args.bundle_merge_distance):
            f.write(line + "\n")
    with open(with_extension(args.output_prefix, "ctg.summary.tsv"), "w") as f:
        for line in pbundle_summary_lines(seq_info, parts, args.k):
            f.write(line + "\n")
    print("%d sequences, %d principal bundles -> %s.bed / .ctg.summary.tsv" % (len(seq_info), len(bundles),
                                                                               args.output_prefix), file=sys.stderr)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="pgrtk_amd.cli")
    sub = ap.add_subparsers(dest="cmd", required=True)
    m = sub.add_parser("mdb", help="pgr-mdb counterpart: FASTA list -> .mdb/.midx")
    m.add_argument("filepath")
    m.add_argument("prefix", nargs="?", default=None)
    m.add_argument("--synthetic", default=None, metavar="NxL",
                   help="N contigs of L bases from the counter-based generator (generated on the device); no <filelist>")
    m.add_argument("--seed", type=int, default=0)
    m.add_argument("--write-fasta", dest="write_fasta", default=None, help="with --synthetic: also write the contigs as FASTA")
    m.add_argument("-w", type=int, default=80)
    m.add_argument("-k", type=int, default=56)
    m.add_argument("-r", type=int, default=4)
    m.add_argument("-m", "--min-span", dest="min_span", type=int, default=64)
    m.add_argument("--sketch", action="store_true")
    m.add_argument("--batch-bp", type=int, default=2_000_000_000)
    m.add_argument("--reference-sid-quirk", action="store_true",
                   help="restart sequence ids at 0 for every input file, like load_index_from_reader (seq_db.rs:543)")
    m.set_defaults(fn=cmd_mdb)
    q = sub.add_parser("query", help="pgr-query counterpart")
    q.add_argument("pgr_db_prefix")
    q.add_argument("query_fastx_path")
    q.add_argument("output_prefix")
    # (clap 4 derives kebab-case long names, pgr-query.rs:26-30; the snake_case spelling of earlier rounds stays accepted)
    q.add_argument("--fastx-file", "--fastx_file", dest="fastx_file", action="store_true")
    q.add_argument("--frg-file", "--frg_file", dest="frg_file", action="store_true",
                   help="the reference's .frg sequence store: not supported (use the .mdb/.midx pair or --fastx-file)")
    q.add_argument("-w", type=int, default=80)
    q.add_argument("-k", type=int, default=56)
    q.add_argument("-r", type=int, default=4)
    q.add_argument("-m", "--min-span", dest="min_span", type=int, default=64)
    q.add_argument("-g", "--gap-penalty-factor", dest="gap_penalty_factor", type=float, default=0.025)
    q.add_argument("--merge-range-tol", dest="merge_range_tol", type=int, default=100000)
    q.add_argument("--max-count", dest="max_count", type=int, default=128)
    q.add_argument("--max-query-count", dest="max_query_count", type=int, default=128)
    q.add_argument("--max-target-count", dest="max_target_count", type=int, default=128)
    q.add_argument("--max-aln-chain-span", dest="max_aln_chain_span", type=int, default=8)
    q.add_argument("--only-summary", dest="only_summary", action="store_true")
    q.add_argument("--bed-summary", dest="bed_summary", action="store_true")
    q.set_defaults(fn=cmd_query)
    b = sub.add_parser("pbundle-decomp", help="pgr-pbundle-decomp counterpart")
    b.add_argument("fastx_path")
    b.add_argument("output_prefix")
    b.add_argument("-i", "--include", default=None)
    b.add_argument("-d", "--decomp-fastx-path", dest="decomp_fastx_path", default=None)
    b.add_argument("--precomputed-bundles", dest="precomputed_bundles", default=None,
                   help="a .pdb written by an earlier run: skip the bundle computation (its w/k/r/min_span/... win)")
    b.add_argument("-w", type=int, default=48)
    b.add_argument("-k", type=int, default=56)
    b.add_argument("-r", type=int, default=4)
    b.add_argument("--min-span", dest="min_span", type=int, default=12)
    b.add_argument("--min-cov", dest="min_cov", type=int, default=0)
    b.add_argument("--min-branch-size", dest="min_branch_size", type=int, default=8)
    b.add_argument("--bundle-length-cutoff", dest="bundle_length_cutoff", type=int, default=2500)
    b.add_argument("--bundle-merge-distance", dest="bundle_merge_distance", type=int, default=10000)
    b.set_defaults(fn=cmd_pbundle_decomp)
    args = ap.parse_args(argv)
    return args.fn(args)


if __name__ == "__main__":
    sys.exit(main() or 0)
