set -x
python -m pytest tests -m gpu -x -q -k "file_api or c_driver or reference_test_data or handmade" > gpurun_out/r03j_tests.log 2>&1; tail -3 gpurun_out/r03j_tests.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
$B --opt gather_unroll=1 > gpurun_out/r03j_u1.json 2> gpurun_out/r03j.err
$B --opt gather_unroll=4 > gpurun_out/r03j_u4.json 2>> gpurun_out/r03j.err
$B --opt gather_unroll=1 > gpurun_out/r03j_u1b.json 2>> gpurun_out/r03j.err
timeout -k 10 500 python tools/file_api_times.py --contigs 3000000 > gpurun_out/r03j_file_api_3M.json 2>> gpurun_out/r03j.err
tail -3 gpurun_out/r03j.err
