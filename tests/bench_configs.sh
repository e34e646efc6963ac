#!/bin/bash
# All five BASELINE.json configs at full size on one GPU (one step each = the whole render); prints one JSON line per config.
for w in c1_default_256_1spp_b1 cornell_512_64spp_b8 c3_cornell2_1080p_ssaa2_1024spp_b16 c4_cornell2_2160p_1024spp_b16 c5_mesh_1080p_512spp c5_minecraft_1080p_ssaa2_512spp; do
  python bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline
done
