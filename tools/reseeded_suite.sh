#!/bin/bash
# On the GPU box: the GPU parity suite once per seed shift (tests/conftest.py, PQD_SEED_SHIFT), one process each, stopping
# at the first run that fails.  Output: gpurun_out/seed/shift<n>.txt
set -e
mkdir -p gpurun_out/seed
for s in "$@"; do
  echo "shift $s"
  PQD_SEED_SHIFT=$s timeout -k 10 600 python3 -u -m pytest tests -q -m gpu --timeout 120 --timeout-method thread -p no:cacheprovider > gpurun_out/seed/shift$s.txt 2>&1 || { tail -30 gpurun_out/seed/shift$s.txt; exit 1; }
  tail -1 gpurun_out/seed/shift$s.txt
done
