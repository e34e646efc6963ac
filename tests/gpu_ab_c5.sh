set -o pipefail
cd $GRAFT_REPO_ROOT
run() { name=$1; wl=$2; shift; shift
  env "$@" timeout -k 10 300 python bench.py --workload $wl --steps ${K:-8} --warmup 3 --no-cpu-baseline 2>>gpurun_out/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernel'])"
}
for i in 1 2; do
run "mesh5k deep " mesh5k_1080p_64spp
run "mesh5k L2   " mesh5k_1080p_64spp MRT_SCENE_IN_L2=1
run "mesh20k deep" mesh20k_540p_64spp
run "mesh20k L2  " mesh20k_540p_64spp MRT_SCENE_IN_L2=1
done
K=2 run "minecraft warm6" c5_minecraft_1080p_ssaa2_512spp
K=2 run "minecraft all  " c5_minecraft_1080p_ssaa2_512spp MRT_COLD=0
K=3 run "mesh967 warm q " c5_mesh_1080p_512spp
K=3 run "mesh967 all    " c5_mesh_1080p_512spp MRT_COLD=0
