"""How many pool threads, and bound to the GPU's NUMA node or not?  15 pgr_shmmr_batch calls (104 x 10 Mbp ASCII, host in / host
out) per setting in a fresh process each: median / min / max.  The GPU boxes grant 16 CPUs' worth of time (cgroup quota) on 256
visible CPUs: more runnable threads than the quota means throttled periods."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [os.path.join(ROOT, "pgr-tk_amd"), ROOT]
    import bench
    import pgrtk_amd as P
    ctx = P.default_context(0)
    seqs = [bench.synth_contig_ascii(2, i, 10_000_000) for i in range(104)]
    sp = P.make_spec()
    packed, _ = P.pack_ascii(seqs)
    bare = P.PackedBases(packed.lens, packed.planes, None)
    for name, f in (("ascii", lambda: P.time_shmmr_batch(seqs, sp, ctx=ctx)[0]), ("planes", lambda: P.time_shmmr_batch_packed(bare, sp, ctx=ctx)[0])):
        f()
        ts = sorted(f() for _ in range(15))
        print("  %-6s median %6.2f ms  min %6.2f  max %6.2f  -> %5.1f Gbp/s (median)" % (name, ts[7] * 1e3, ts[0] * 1e3, ts[-1] * 1e3, 1.04 / ts[7]))
    thr = open("/sys/fs/cgroup/cpu.stat").read().split()
    print("  cgroup: nr_throttled %s throttled_usec %s" % (thr[thr.index("nr_throttled") + 1], thr[thr.index("throttled_usec") + 1]))
    sys.exit(0)
for threads in (8, 10, 12, 14, 16):
    for bind in (1, 0):
        env = dict(os.environ, PGR_HOST_THREADS=str(threads))
        if not bind:
            env["PGR_NO_NUMA_BIND"] = "1"
        print("threads %d, %s" % (threads, "bound to the GPU's node" if bind else "not bound"), flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, stderr=subprocess.DEVNULL)
