#!/bin/bash
# Collect the rocprofv3 evidence of one bench workload on the GPU box (run through gpurun from the repo root):
#   profiles/collect.sh <tag> [workload] [extra bench.py flags for the PMC passes]
# Kernel trace + stats of the bench command, then separate --pmc passes (never combined with tracing), all under
# gpurun_out/<tag>_*; `python profiles/summarize.py <tag> <workload>` turns them into profiles/<tag>_*.
set -e
tag=$1; wl=${2:-cornell_1080p_1024spp_b8}; shift; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_trace -o out --output-format csv -- python3 bench.py --workload $wl --no-cpu-baseline "$@" > gpurun_out/${tag}_trace.log 2>&1
pass() { name=$1; shift; rocprofv3 --pmc "$@" -d gpurun_out/${tag}_pmc_$name -o out --output-format csv -- python3 bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline $EXTRA > gpurun_out/${tag}_pmc_$name.log 2>&1; }
pass sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pass wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
pass mix SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FLOPS_FP32
pass fetch FETCH_SIZE
pass write WRITE_SIZE
tail -1 gpurun_out/${tag}_trace.log | cut -c1-300
