set -x
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r03l_head.json 2> gpurun_out/r03l.err
timeout -k 10 500 python tools/file_api_times.py --contigs 3000000 > gpurun_out/r03l_file_api_3M.json 2>> gpurun_out/r03l.err
tail -3 gpurun_out/r03l.err
