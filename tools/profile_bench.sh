#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   tools/profile_bench.sh <tag>      -> gpurun_out/prof_<tag>/{trace,pmc_fetch,pmc_write,pmc_sq,pmc_misc,q_trace,q_pmc_*}
# Kernel trace/stats and every PMC group are SEPARATE runs (never --pmc together with tracing domains).
set -u
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --queries 0 --no-extras --no-pipelined-leg"   # only full-size launches of every kernel (the first launches run at ramping clocks: enough steps for the average to mean something)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- $B > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- $B > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY \
          --output-format csv -d $O/pmc_sq -o p -- $B > $O/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
          --output-format csv -d $O/pmc_misc -o p -- $B > $O/pmc_misc.log 2>&1
# VALU busy by counter (round 3): cycles with a VALU instruction active per SIMD against the cycles the shader engines were busy
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_CYCLES \
          --output-format csv -d $O/pmc_valu -o p -- $B > $O/pmc_valu.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE \
          --output-format csv -d $O/pmc_valu2 -o p -- $B > $O/pmc_valu2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/q_trace -o q -- python $R/tools/query_leg.py --reps 5 > $O/q_trace.log 2>&1
# the query leg's counters (separate passes as well): HBM bytes and VALU work per query batch
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/q_pmc_fetch -o q -- python $R/tools/query_leg.py --reps 5 > $O/q_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/q_pmc_write -o q -- python $R/tools/query_leg.py --reps 5 > $O/q_pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_LDS --output-format csv -d $O/q_pmc_sq -o q -- python $R/tools/query_leg.py --reps 5 > $O/q_pmc_sq.log 2>&1
# the software pipeline (pgr_pipe_*): kernel trace of the pipelined leg alone -- which kernels run beside the tile kernel, and what it costs it
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pipe_trace -o p -- python $R/tools/probe/pipe_probe.py 1000 8 > $O/pipe_trace.log 2>&1
rm -f $O/pipe_trace/*kernel_trace.csv  # (the per-dispatch trace: only the stats travel back)
python $R/bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err
ls -R $O | head -40
