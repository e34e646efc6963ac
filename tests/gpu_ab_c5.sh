# scratch: list SQC counters and collect instruction-cache / fetch counters on the mesh workload
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|SQC_|SQ_WAIT_IFETCH|SQ_INST_LEVEL|SQ_WAVE_DEP|SQ_WAIT_INST" | cut -c1-160 | sort -u > gpurun_out/counters.txt; wc -l gpurun_out/counters.txt; head -60 gpurun_out/counters.txt
NOCOOP=$GRAFT_REPO_ROOT/micro_raytracer_amd/libmrt_hip_nocoop.so
pass() { name=$1; shift; MRT_LIB=$NOCOOP rocprofv3 --pmc "$@" -d gpurun_out/ic_$name -o out --output-format csv -- python3 bench.py --workload c5_mesh_1080p_512spp --steps 1 --warmup 0 --spp 64 --no-cpu-baseline > gpurun_out/ic_$name.log 2>&1; }
pass a SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass b SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/ic_*/**/*counter_collection.csv", recursive=True)):
    one = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "pt_megakernel" in r["Kernel_Name"]:
            one[r["Counter_Name"]] += float(r["Counter_Value"])
    print(f.split('/')[1], dict(one))
PY
