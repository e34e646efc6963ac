#!/bin/bash
# Every BASELINE.json config at full size on one GPU (one step = the whole render), the per-call workloads and the meshes
# beyond the LDS; one JSON line per workload, cpu_baseline included (profiles/r3_configs.jsonl).  Renders shorter than
# ~100 ms are timed over 10 steps after 3 warm-up steps (clocks, L2), the long ones over one.
#   bash tests/bench_configs.sh [workload ...]      (default: all)
ALL="c1_default_256_1spp_b1 cornell_512_64spp_b8 cornell_1080p_1024spp_b8 c3_cornell2_1080p_ssaa2_1024spp_b16 c4_cornell2_2160p_1024spp_b16 c5_mesh_1080p_512spp c5_minecraft_1080p_ssaa2_512spp cornell_1080p_percall cornell_1080p_percall_deferred mesh5k_1080p_64spp mesh20k_540p_64spp"
for w in ${@:-$ALL}; do
  case $w in
    c1_default_256_1spp_b1|cornell_512_64spp_b8|mesh5k_1080p_64spp|mesh20k_540p_64spp) k="--steps 10 --warmup 3" ;;
    *) k="--steps 1 --warmup 1" ;;
  esac
  python bench.py --workload $w $k
done
