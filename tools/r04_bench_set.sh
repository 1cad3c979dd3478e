#!/bin/bash
# Round-4 record set (run on the GPU box from the repo root): bench lines of every operating point, latency at both
# visibility volumes, the kernel stats of the default bench (every FIM worker instantiation by name), the counter passes at the
# reference's visibility request, and the 2-rank gloo rehearsal of the N > 1 line.  Outputs under gpurun_out/r04/.
set -u
O=gpurun_out/r04/final
mkdir -p $O
PY=$(python3 -c "import os,sys;print(os.path.realpath(sys.executable))")
run() { name=$1; shift; timeout -k 10 400 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; }
run c3_bench python3 bench.py
run c3_ref_request_bench python3 bench.py --fim-angle 4.0
run ref2d_bench python3 bench.py --workload REF2D --cpu-seconds 4
run c5_bench python3 bench.py --workload C5 --cpu-seconds 0 --no-parity
run c3_l160_bench python3 bench.py --depth-cells 160 --cpu-seconds 0 --no-parity
run c4_strong_n1_bench python3 bench.py --scaling strong --cpu-seconds 0 --no-parity
run latency_operating_point python3 bench.py --latency --latency-calls 500
run latency_reference_request python3 bench.py --latency --latency-calls 500 --fim-angle 4.0
run gloo2_rehearsal_bench python3 bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --repeats 3
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_default -- $PY $GRAFT_REPO_ROOT/bench.py --steps 20 --cpu-seconds 0 > $GRAFT_REPO_ROOT/$O/stats_default.log 2>&1; echo "stats rc=$?" )
cp $(find $O/stats_default -name "*kernel_stats.csv" | head -1) $O/c3_all_workers_kernel_stats.csv 2>/dev/null
timeout -k 10 500 python3 tools/pmc_collect.py --out gpurun_out/pmc_r04_ref --passes insts,cycles,stalls,grbm,fetch,write,tcc --bench-arg=--fim-angle --bench-arg=4.0 > $O/pmc_ref.log 2>&1; echo "pmc_ref rc=$?"
