#!/bin/bash
# Timing-only ablations of the FIM worker on the GPU box (results of the ablated builds are wrong by construction):
# builds the library with FS_FIM_ABLATE = 3, 2, 1 and prints the kernel times of bench.py for each, then restores
# the production build.  Usage (from the repo root): bash tools/fim_ablate.sh gpurun_out/ablate
set -e
out=${1:-gpurun_out/ablate}; mkdir -p "$out"
for n in 3 2 1; do
  FS_FIM_ABLATE=$n python fit-slam_amd/_build.py --force > "$out/build_$n.log" 2>&1
  python bench.py --steps 10 --warmup 3 --repeats 3 --cpu-seconds 0 --no-parity > "$out/bench_$n.json" 2> "$out/bench_$n.err"
  python - "$out/bench_$n.json" $n <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); print("ablate", sys.argv[2], "ms/step %.4f" % j["ms_per_step"], j["kernels_ms_per_step"])
PY
done
python fit-slam_amd/_build.py --force > "$out/build_0.log" 2>&1
python bench.py --steps 10 --warmup 3 --repeats 3 --cpu-seconds 0 --no-parity > "$out/bench_0.json" 2> "$out/bench_0.err"
python - "$out/bench_0.json" 0 <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); print("ablate", sys.argv[2], "ms/step %.4f" % j["ms_per_step"], j["kernels_ms_per_step"])
PY
