set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; wl=$2; spp=$3
pass() { name=$1; shift; rocprofv3 --pmc "$@" -d gpurun_out/${tag}_pmc_$name -o out --output-format csv -- python3 bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --spp $spp > gpurun_out/${tag}_pmc_$name.log 2>&1; }
pass sq SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pass wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
python3 - <<PY
import csv, glob, collections
for name in ("sq","wait"):
    for f in glob.glob("gpurun_out/${tag}_pmc_%s/**/*counter_collection.csv" % name, recursive=True):
        one = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "pt_megakernel" in r["Kernel_Name"]:
                one[r["Counter_Name"]] += float(r["Counter_Value"]); meta=(r["VGPR_Count"], r["Scratch_Size"], r["Kernel_Name"][:60])
        print(name, meta, dict(one))
PY
