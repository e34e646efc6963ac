set -o pipefail
cd $GRAFT_REPO_ROOT
run() { name=$1; wl=$2; shift; shift
  env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline 2>>gpurun_out/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value']), d['roofline']['kernel'], d['kernel'])"
}
run "minecraft " c5_minecraft_1080p_ssaa2_512spp
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "scene_parity or launch_shape or staging" > gpurun_out/ab_tests.log 2>&1; echo "rc $?" >> gpurun_out/ab_tests.log; tail -3 gpurun_out/ab_tests.log
