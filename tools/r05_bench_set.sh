#!/bin/bash
# Round-5 record set (run on the GPU box from the repo root): the bench lines of every operating point on the final sources
# (with the one tolerance rule, the whole-socket CPU figures and --min-timed-seconds), the one-rank rehearsal of the N > 1 line
# (weak blocks + the strong-scaling block), latency at both visibility volumes, the kernel stats of the default bench under
# rocprofv3, and the counter passes at both volumes.  Outputs under gpurun_out/r05/final.
set -u
O=gpurun_out/r05/final
mkdir -p $O
PY=$(python3 -c "import os,sys;print(os.path.realpath(sys.executable))")
run() { name=$1; shift; timeout -k 10 500 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; }
# counters FIRST: the bench lines below replay profiles/pmc_summary*.json, which must carry the hash of the sources being run
timeout -k 10 900 python3 tools/pmc_collect.py --out gpurun_out/r05/pmc_c3 > $O/pmc_c3.log 2>&1; echo "pmc_c3 rc=$?"
timeout -k 10 600 python3 tools/pmc_collect.py --out gpurun_out/r05/pmc_c3_ref --skip-calib --passes insts,cycles,stalls,grbm,fetch,write,tcc --bench-arg=--fim-angle --bench-arg=4.0 > $O/pmc_ref.log 2>&1; echo "pmc_ref rc=$?"
python3 tools/pmc_collect.py --rederive gpurun_out/r05/pmc_c3_ref --calib-from gpurun_out/r05/pmc_c3/pmc_summary.json > /dev/null 2>&1
cp gpurun_out/r05/pmc_c3/pmc_summary.json profiles/pmc_summary.json 2>/dev/null
cp gpurun_out/r05/pmc_c3_ref/pmc_summary.json profiles/pmc_summary_ref_request.json 2>/dev/null
run c3_bench python3 bench.py
run c3_ref_request_bench python3 bench.py --fim-angle 4.0
run ref2d_bench python3 bench.py --workload REF2D
run c5_bench python3 bench.py --workload C5 --cpu-seconds 6
run c3_l160_bench python3 bench.py --depth-cells 160 --cpu-seconds 0 --no-parity
run c4_strong_n1_bench python3 bench.py --scaling strong --cpu-seconds 0 --no-parity
run rehearse_multi_one_rank_rccl_bench python3 bench.py --rehearse-multi
run latency_operating_point python3 bench.py --latency --latency-calls 500
run latency_reference_request python3 bench.py --latency --latency-calls 500 --fim-angle 4.0
run gloo2_rehearsal_bench python3 bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --repeats 3 --min-timed-seconds 0
run grid_region_update python3 tools/grid_region_probe.py
