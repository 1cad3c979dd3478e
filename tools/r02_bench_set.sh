# the round-2 operating points, one JSON line each under gpurun_out/ (copied into profiles/r02/ afterwards)
set -e
python bench.py > gpurun_out/c3_bench.json 2> gpurun_out/c3_bench.err
python bench.py --workload C5 --cpu-seconds 0 --steps 10 > gpurun_out/c5_single_gpu_bench.json 2> gpurun_out/c5.err
python bench.py --scaling strong --cpu-seconds 0 --steps 10 > gpurun_out/c4_strong_n1_bench.json 2> gpurun_out/c4.err
python bench.py --workload REF2D --cpu-seconds 0 > gpurun_out/ref2d_bench.json 2> gpurun_out/ref2d.err
python bench.py --gpus 2 --backend gloo --cpu-seconds 0 --steps 10 > gpurun_out/gloo2_rehearsal_bench.json 2> gpurun_out/gloo2.err
python bench.py --gpus 2 --backend gloo --scaling strong --cpu-seconds 0 --steps 5 > gpurun_out/gloo2_rehearsal_strong_bench.json 2> gpurun_out/gloo2s.err
python bench.py --depth-cells 160 --cpu-seconds 0 --steps 10 > gpurun_out/c3_l160_bench.json 2> gpurun_out/c3_l160.err
