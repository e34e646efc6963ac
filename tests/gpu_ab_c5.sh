set -o pipefail
cd $GRAFT_REPO_ROOT
run() { name=$1; wl=$2; shift; shift
  env "$@" timeout -k 10 300 python bench.py --workload $wl --steps ${K:-3} --warmup 1 --no-cpu-baseline 2>>gpurun_out/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernel'], d['kernel']['lds_bytes'], d['roofline'].get('reduce_ms'))"
}
K=1 run "percall default (64)" cornell_1080p_percall
K=1 run "percall 256 nopersist" cornell_1080p_percall MRT_BLOCK_THREADS=256 MRT_NO_PERSIST=1
K=2 run "mesh k16" c5_mesh_1080p_512spp MRT_K_SPLIT=16
K=2 run "mesh k8" c5_mesh_1080p_512spp
K=1 run "minecraft k1" c5_minecraft_1080p_ssaa2_512spp
K=1 run "minecraft k2" c5_minecraft_1080p_ssaa2_512spp MRT_K_SPLIT=2
K=1 run "minecraft k4" c5_minecraft_1080p_ssaa2_512spp MRT_K_SPLIT=4
K=3 run "headline nopersist 256" cornell_1080p_1024spp_b8 MRT_NO_PERSIST=1
