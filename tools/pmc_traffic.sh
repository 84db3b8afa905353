#!/bin/bash
# On the GPU box, from the repository root: the two counter passes behind profiles/pmc_traffic.json.
#   gpurun -- 'bash tools/pmc_traffic.sh'   then   python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv profiles/pmc_traffic.json
set -e
export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-verify --no-extras > gpurun_out/pmc_fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-verify --no-extras > gpurun_out/pmc_write.log 2>&1
