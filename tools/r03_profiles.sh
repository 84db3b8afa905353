#!/bin/bash
# On the GPU box, from the repository root: the runs behind profiles/r03_* (each step under its own time limit).
set -e
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 300 python3 -u bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-extras > $O/bench_line_profiled.json 2> $O/trace.err
bash tools/pmc_traffic.sh
timeout -k 10 900 python3 -u tests/perf/kernel_rates.py > $O/kernel_rates.txt 2> $O/kernel_rates.err
KR_NO_SPOILER=1 KR_ONLY="ONE page" timeout -k 10 200 python3 -u tests/perf/kernel_rates.py > $O/one_page_warm.txt 2>&1
KR_ONLY=decompress KR_LZ_PAGES=256 timeout -k 10 400 python3 -u tests/perf/kernel_rates.py > $O/kernel_rates_256_pages.txt 2> $O/kr256.err
KR_ONLY=decompress timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace_dec -o dec -- python3 tests/perf/kernel_rates.py > $O/kr_dec_profiled.txt 2> $O/trace_dec.err
KR_ONLY=decompress KR_LZ_PAGES=256 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace_dec256 -o dec -- python3 tests/perf/kernel_rates.py > $O/kr_dec256_profiled.txt 2> $O/trace_dec256.err
timeout -k 10 300 python3 -u tools/e2e_bench.py > $O/e2e.txt 2>&1
timeout -k 10 600 python3 -u tools/e2e_compressed.py > $O/e2e_compressed.txt 2>&1
timeout -k 10 600 python3 -u tools/e2e_compressed.py --host-threads 16 > $O/e2e_compressed_16_host_threads.txt 2>&1
echo done
