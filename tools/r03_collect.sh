#!/bin/bash
# Here, after tools/r03_profiles.sh has run on the GPU box: gpurun_out/r03 -> profiles/r03_* and profiles/pmc_traffic.json.
set -e
O=gpurun_out/r03
cp $O/bench_line.json profiles/r03_bench_line.json
cp $O/bench_line_profiled.json profiles/r03_bench_line_profiled.json
python tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) profiles/r03_kernel_stats_bench.csv
python tools/rocpd_stats.py $(find $O/trace_dec -name "*.db" | head -1) profiles/r03_kernel_stats_decompress_1792_pages.csv
python tools/rocpd_stats.py $(find $O/trace_dec256 -name "*.db" | head -1) profiles/r03_kernel_stats_decompress_256_pages.csv
python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv profiles/pmc_traffic.json
{
  grep -v "amdgpu.ids" $O/kernel_rates.txt
  echo
  echo "# the decompression cases with 256 pages in flight (KR_LZ_PAGES=256)"
  grep -v "amdgpu.ids" $O/kernel_rates_256_pages.txt
  echo
  echo "# the one-page case with the input left in the caches between iterations (KR_NO_SPOILER=1)"
  grep "ONE page" $O/one_page_warm.txt
} > profiles/r03_kernel_rates.txt
{
  echo "# tools/e2e_bench.py: the bench columns INT32 + INT64 with the PCIe copies included"
  grep -v "amdgpu.ids" $O/e2e.txt
  echo
  echo "# tools/e2e_compressed.py: one column chunk, compressed bytes up / values down, against pages decompressed on the host"
  grep -v "amdgpu.ids" $O/e2e_compressed.txt
} > profiles/r03_end_to_end.txt
