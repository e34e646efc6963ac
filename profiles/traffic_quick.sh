set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 300 rocprofv3 --pmc $c -d gpurun_out/r4p_$c -o out --output-format csv -- python3 bench.py --workload c5_minecraft_1080p_ssaa2_512spp --steps 1 --warmup 0 --no-cpu-baseline --spp 32 > gpurun_out/r4p_$c.log 2>&1; done
python3 - <<PY
import csv, glob
for c in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob("gpurun_out/r4p_%s/**/*counter_collection.csv" % c, recursive=True):
        t=0
        for r in csv.DictReader(open(f)):
            if "pt_megakernel" in r["Kernel_Name"]: t+=float(r["Counter_Value"]); sc=r["Scratch_Size"]
        print(c, t*1024/1e6, "MB (x2 for FETCH)", "scratch", sc)
PY
