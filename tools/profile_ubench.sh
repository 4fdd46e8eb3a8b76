#!/bin/bash
# The VALU micro-benchmarks under rocprofv3 (run through gpurun):  tools/profile_ubench.sh <tag>
#   -> gpurun_out/ubench_<tag>/{ub_pmc, ub_pmc_busy, ub_trace, ubench_valu.txt};  condensed by tools/ubench_table.py
set -u
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ubench_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
(cd $R/tools && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 ubench_valu.hip -o ubench_valu) > $O/build.log 2>&1
$R/tools/ubench_valu > $O/ubench_valu.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/ub_pmc -o p -- $R/tools/ubench_valu > $O/ub_pmc.log 2>&1
# calibration of the VALU-busy counters: these kernels keep the VALU issue port busy by construction
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_CYCLES --output-format csv -d $O/ub_pmc_busy -o p -- $R/tools/ubench_valu > $O/ub_pmc_busy.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ub_trace -o p -- $R/tools/ubench_valu > $O/ub_trace.log 2>&1
ls -R $O | head -30
