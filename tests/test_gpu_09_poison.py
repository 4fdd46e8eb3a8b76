"""debug_poison: lanes, cached blocks and workspaces are reused from call to call; a kernel that reads what an
EARLIER call left in one of them works or faults depending on the layout of the moment (round 5 found one such read only by
fuzzing, profiles/r05_fuzz/cursor_block_size_fault.txt).  With the context option debug_poison every block that is about to be used
again is filled with 0xFF first and the list stage checks the segment table it is about to read, so that such a read fails the call
-- deterministically, in the first case that takes the path.

  * the defect of round 5, put back by fault injection (option debug_inject_stale_segments), is caught in its FIRST pipelined job;
  * the pipe's parity tests (tests/test_gpu_07_pipe.py) and 200 cases of tools/fuzz_pipe.py pass with the option on.
"""
import os
import re
import sys

import numpy as np
import pytest

import procutil
import seqgen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pipe_lists(P, ctx, spec, batches):
    pipe = P.Pipe(spec, ctx=ctx)
    got = []
    try:
        for b in batches:
            if pipe.in_flight == 2:
                got.append(pipe.collect()[0])
            pipe.submit(b)
        while pipe.in_flight:
            got.append(pipe.collect()[0])
    finally:
        pipe.close()
    return got


def test_poison_catches_the_stale_segment_table_in_the_first_job(oracle):
    """Jobs of a w = 80 spec leave their segment counts in the lanes; then a w = 9 spec (no tile kernel) runs on the same lanes with
    the list stage in its optimistic pass (no_stage1_only) and WITHOUT the clearing memset (debug_inject_stale_segments = the tree
    before round 5's fix).  Without debug_poison that is a wild read that may or may not fault; with it the first collect fails with
    the library's message.  The same jobs without the injection: bit exact against the oracle, poison on."""
    import pgrtk_amd as P
    ctx = P.Context(0)
    ctx.set_option("debug_poison", 1)
    rng = np.random.default_rng(77)
    big = [P.Batch.from_seqs([seqgen.rnd(rng, 1_200_000), seqgen.rnd(rng, 700_000)], ctx=ctx) for _ in range(3)]
    _pipe_lists(P, ctx, P.make_spec(80, 56, 4, 64, False), big)
    sets = [[seqgen.rnd(rng, int(L)) for L in rng.integers(20_000, 300_000, 4)] for _ in range(3)]
    batches = [P.Batch.from_seqs(s, ctx=ctx) for s in sets]
    spec_t = (9, 12, 3, 8, False)
    spec, osp = P.make_spec(*spec_t), oracle.spec(*spec_t)
    with ctx.options(no_stage1_only=1, debug_inject_stale_segments=1):
        with pytest.raises(P.PgrError) as ei:
            _pipe_lists(P, ctx, spec, batches)
    assert "debug_poison" in str(ei.value) and "segment table" in str(ei.value), str(ei.value)
    for opts in ({"no_stage1_only": 1}, {}):
        with ctx.options(**opts):
            got = _pipe_lists(P, ctx, spec, batches)
        for bi, sh in enumerate(got):
            mm, off = sh.download()
            for i, q in enumerate(sets[bi]):
                ref = oracle.sequence_to_shmmrs(i, q, osp)
                g = mm[int(off[i]):int(off[i + 1])]
                assert len(ref) == len(g) and np.array_equal(ref["x"], g["x"]) and np.array_equal(ref["y"], g["y"]), (opts, bi, i)


def test_pipe_parity_tests_pass_with_poison_on():
    """tests/test_gpu_07_pipe.py once more in a process whose contexts are created with PGR_DEBUG_POISON=1"""
    r = procutil.run_bounded([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_07_pipe.py"), "-x", "-q", "-m", "gpu",
                              "-p", "no:cacheprovider"], timeout=400, env={"PGR_DEBUG_POISON": "1"}, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 12 and "failed" not in r.stdout, r.stdout[-800:]


def test_fuzz_pipe_200_cases_with_poison_on():
    """200 cases of the pipe's randomised campaign (random specs, adversarial batches, direct / staged records, flagged batches that
    finish on the fix stream) under PGR_DEBUG_POISON=1: bit exact against the oracle, no call fails"""
    r = procutil.run_bounded([sys.executable, os.path.join(ROOT, "tools", "fuzz_pipe.py"), "200", "61000", "250000"], timeout=400,
                             env={"PGR_DEBUG_POISON": "1"}, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert re.search(r"\b0 failures", r.stdout) or re.search(r"failures[:=]? *0\b", r.stdout), r.stdout[-800:]


def test_first_commit_into_a_fresh_append_block_is_ordered_behind_the_block_s_fill():
    """Round 6, found by 6 000 fuzz_pipe cases under poison (seeds 9200023 and 9204253; timing dependent: 4 runs in 12): the first commit
    of a pipe's records into a FRESH append block of an index (csrc/index.hip: index_grow_raw with no old records to copy) waited for
    nothing -- the block had been ordered, and under debug_poison filled, on the context's stream, the commit copy ran on the back
    stream, and the fill landed on top of the copy: a job's records replaced by 0xFF in the finalized index.  (Without the fill the same
    gap lets a block's previous life race its first writer on another stream.)  index_grow_raw now copies on the stream the allocator
    was asked for and always waits for it.  The 24 cases that led up to the first failing seed, four times over."""
    for _ in range(4):
        r = procutil.run_bounded([sys.executable, os.path.join(ROOT, "tools", "fuzz_pipe.py"), "24", "9200000"], timeout=200,
                                 env={"PGR_DEBUG_POISON": "1"})
        assert r.returncode == 0 and "0 failures" in r.stdout, (r.stdout[-1500:], r.stderr[-500:])
