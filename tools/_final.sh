set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r03zg_tests.log 2>&1; tail -2 gpurun_out/r03zg_tests.log
B="python bench.py --inversions 0.1 --duplicate-pairs --steps 5 --warmup 2 --no-cpu-baseline --no-secondary"
$B > gpurun_out/r03zg_inv_on_a.json 2> gpurun_out/r03zg.err
$B --opt defer_unclean_work=0 > gpurun_out/r03zg_inv_off_a.json 2>> gpurun_out/r03zg.err
$B > gpurun_out/r03zg_inv_on_b.json 2>> gpurun_out/r03zg.err
$B --opt defer_unclean_work=0 > gpurun_out/r03zg_inv_off_b.json 2>> gpurun_out/r03zg.err
bash tools/opt_ab.sh r03zg_head defer_unclean_work=0
timeout -k 10 400 python tools/file_api_times.py --contigs 3000000 > gpurun_out/r03zg_file_api_3M.json 2> gpurun_out/r03zg_file_api.err
tail -2 gpurun_out/r03zg_file_api.err
