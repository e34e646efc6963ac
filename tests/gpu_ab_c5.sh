set -o pipefail
cd $GRAFT_REPO_ROOT
run() { name=$1; wl=$2; shift; shift
  env "$@" timeout -k 10 300 python bench.py --workload $wl --steps ${K:-3} --warmup 1 --no-cpu-baseline 2>>gpurun_out/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernel'], d['kernel']['lds_bytes'], d['roofline'].get('reduce_ms'))"
}
run "headline" cornell_1080p_1024spp_b8
K=1 run "c3" c3_cornell2_1080p_ssaa2_1024spp_b16
K=1 run "c4" c4_cornell2_2160p_1024spp_b16
K=10 run "cornell512" cornell_512_64spp_b8
K=1 run "percall" cornell_1080p_percall
K=1 run "deferred" cornell_1080p_percall_deferred
K=2 run "mesh" c5_mesh_1080p_512spp
K=1 run "minecraft" c5_minecraft_1080p_ssaa2_512spp
K=20 run "c1" c1_default_256_1spp_b1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/ab_tests.log 2>&1; echo "rc $?" >> gpurun_out/ab_tests.log; tail -3 gpurun_out/ab_tests.log
