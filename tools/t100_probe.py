import sys, os, time, json
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"), "pgr-tk_amd")); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
import bench, pgrtk_amd as P
class A: seed=2
ctx=P.default_context(0); spec=P.make_spec()
r=bench.target_100gbp(P,ctx,spec,A); print("fresh", r["s"], r["batches_s"], r["sort_into_frag_map_s"], flush=True)
r=bench.target_100gbp(P,ctx,spec,A); print("again", r["s"], r["batches_s"], r["sort_into_frag_map_s"], flush=True)
if len(sys.argv)>1:
    s=bench.shapes_bench(P,ctx,spec,(80,56,4,64),16,False); print({k:round(v["Gbp_per_s"],1) for k,v in s.items()}, flush=True)
    r=bench.target_100gbp(P,ctx,spec,A); print("after shapes", r["s"], r["batches_s"], r["sort_into_frag_map_s"], flush=True)
