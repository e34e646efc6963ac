#!/bin/bash
# Quick PMC look at one bench workload (run through gpurun from the repo root): profiles/pmc_quick.sh <tag> <workload> <spp> [set]
# set: sq (default: SQ instruction / wait counters, two passes) | mem (TA / TCP / TCC, three passes).  Prints one dict per pass.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; wl=$2; spp=$3; which=${4:-sq}
pass() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" -d gpurun_out/${tag}_pmc_$name -o out --output-format csv -- python3 bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --spp $spp > gpurun_out/${tag}_pmc_$name.log 2>&1; }
if [ $which = sq ]; then
pass sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pass wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
names="sq wait"
else
pass sqm SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
# (at most two counters of one block per pass: five TA counters at once made rocprofv3 abort and hang, round 4)
pass ta TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
pass tcc TCC_HIT_sum TCC_MISS_sum
names="sqm ta tcp tcc"
fi
python3 - <<PY
import csv, glob, collections
for name in "$names".split():
    for f in glob.glob("gpurun_out/${tag}_pmc_%s/**/*counter_collection.csv" % name, recursive=True):
        one = collections.defaultdict(float); meta = None
        for r in csv.DictReader(open(f)):
            if "pt_megakernel" in r["Kernel_Name"]:
                one[r["Counter_Name"]] += float(r["Counter_Value"]); meta=(r["VGPR_Count"], r["Scratch_Size"], r["Kernel_Name"][:60])
        print(name, meta, dict(one))
PY
