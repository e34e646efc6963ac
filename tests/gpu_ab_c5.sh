set -o pipefail
cd $GRAFT_REPO_ROOT
run() { name=$1; wl=$2; shift; shift
  env "$@" timeout -k 10 300 python bench.py --workload $wl --steps ${K:-3} --warmup 1 --no-cpu-baseline 2>>gpurun_out/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value']), round(d['ms_per_step'],2), d['roofline']['kernel'], d['kernel']['lds_bytes'], d['roofline'].get('reduce_ms'))"
}
run "headline" cornell_1080p_1024spp_b8
run "headline k16" cornell_1080p_1024spp_b8 MRT_K_SPLIT=16
K=1 run "c3" c3_cornell2_1080p_ssaa2_1024spp_b16
K=1 run "deferred" cornell_1080p_percall_deferred
