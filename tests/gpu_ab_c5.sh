set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; wl=$2; shift; shift
  env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline 2>>gpurun_out/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value']), d['roofline']['kernel'], d['kernel'])"
}
run "mesh 967 (warm+queue)" c5_mesh_1080p_512spp
run "mesh 967 all hot     " c5_mesh_1080p_512spp MRT_COLD=0
for c in 1 0; do
MRT_COLD=$c rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY -d gpurun_out/q_$c -o out --output-format csv -- python3 bench.py --workload c5_mesh_1080p_512spp --steps 1 --warmup 0 --spp 64 --no-cpu-baseline > gpurun_out/q_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/q_*/**/*counter_collection.csv", recursive=True)):
    one = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "pt_megakernel" in r["Kernel_Name"]:
            one[r["Counter_Name"]] += float(r["Counter_Value"])
    print(f.split('/')[1], {k: f"{v:.4g}" for k, v in one.items()}, "lane util %.3f" % (one["SQ_THREAD_CYCLES_VALU"]/(64*one["SQ_ACTIVE_INST_VALU"])))
PY
