#!/bin/bash
# The oracle's C / C++ restatement under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only; no GPU involved): a build of the
# same sources into /tmp, loaded by the oracle tests through FSO_ORACLE_SO.  The checker decides parity, so undefined behaviour in
# it would be undefined behaviour in every gate.  Run from the repo root:  bash tools/oracle_sanitize.sh
set -eu
cd "$(dirname "$0")/../oracle"
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
g++ $SAN -std=c++17 -fPIC -ffp-contract=off -c fso_frontier.cpp -o /tmp/fso_frontier_asan.o
gcc $SAN -std=gnu11 -fPIC -ffp-contract=off -mfma -mavx2 -fopenmp fso_raycast.c fso_fisher.c /tmp/fso_frontier_asan.o \
    -o /tmp/libfso_oracle_asan.so -shared -fopenmp -lm -lstdc++
cd ..
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" FSO_ORACLE_SO=/tmp/libfso_oracle_asan.so \
python -m pytest tests/test_oracle_raycast.py tests/test_oracle_fisher.py tests/test_oracle_frontier.py tests/test_oracle_keyframes.py \
    tests/test_oracle_known_answers.py tests/test_golden.py tests/test_reference_held_inputs.py -x -q -m "not gpu" -p no:cacheprovider
