#!/usr/bin/env bash
# CPU-only sanitizer pass (GPU sanitizers are not available on the pool): UBSan builds of the
# oracle and of the host instantiation of csrc/q2048_core*.hpp, then the suites that use them.
set -e
cd "$(dirname "$0")/.."
SAN="-fsanitize=undefined -fno-sanitize-recover=undefined"
gcc -O1 -g -fPIC -std=c11 -ffp-contract=off $SAN -shared -o /tmp/liboracle_ubsan.so oracle/q2048_oracle.c -lm -lpthread
g++ -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off $SAN -I 2048_q-learning_amd/csrc -o /tmp/libhostcheck_ubsan.so tests/hostcheck/hostcheck.cpp
python -c "from oracle import oracle; oracle.build()"; python -m pytest tests/test_core_host.py -q -k luts >/dev/null
cp oracle/liboracle.so /tmp/liboracle_keep.so; cp tests/hostcheck/libhostcheck.so /tmp/libhostcheck_keep.so
trap 'cp /tmp/liboracle_keep.so oracle/liboracle.so; cp /tmp/libhostcheck_keep.so tests/hostcheck/libhostcheck.so; touch oracle/liboracle.so tests/hostcheck/libhostcheck.so' EXIT
cp /tmp/liboracle_ubsan.so oracle/liboracle.so; cp /tmp/libhostcheck_ubsan.so tests/hostcheck/libhostcheck.so
touch oracle/liboracle.so tests/hostcheck/libhostcheck.so
python -m pytest tests/test_oracle_golden.py tests/test_core_host.py -q
