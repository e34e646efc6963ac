#!/bin/bash
# Round-4 evidence in one gpurun call: collect.sh for the headline, the two C5 workloads and the two big-mesh workloads.
# PMC passes of the long C5 launches run at reduced spp (EXTRA), instruction totals are scaled back by summarize.py.
tag=${1:-r4a}
bash profiles/collect.sh ${tag} cornell_1080p_1024spp_b8 && echo "headline done" >> gpurun_out/${tag}_progress.log
EXTRA="--spp 64" bash profiles/collect.sh ${tag}_mesh c5_mesh_1080p_512spp && echo "mesh done" >> gpurun_out/${tag}_progress.log
EXTRA="--spp 32" bash profiles/collect.sh ${tag}_minecraft c5_minecraft_1080p_ssaa2_512spp && echo "minecraft done" >> gpurun_out/${tag}_progress.log
bash profiles/collect.sh ${tag}_mesh5k mesh5k_1080p_64spp --steps 8 --warmup 3 && echo "mesh5k done" >> gpurun_out/${tag}_progress.log
bash profiles/collect.sh ${tag}_mesh20k mesh20k_540p_64spp --steps 8 --warmup 3 && echo "mesh20k done" >> gpurun_out/${tag}_progress.log
