"""host-side timeline (context options debug_times + debug) of one pgr_shmmrs_compute over the repeat-rich contigs of tools/repeat_like_bench.py"""
import os
import sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
src = open(os.path.join(ROOT, "tools", "repeat_like_bench.py")).read().split("sums, off = sh.checksum()")[0]
src = src.replace("ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))", "ROOT = %r" % ROOT)
sys.argv = ["x"]
exec(src)
with ctx.options(debug_times=1, debug=1):  # noqa: F821
    b.shmmrs(sp)  # noqa: F821
