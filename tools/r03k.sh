set -x
python -m pytest tests -m gpu -x -q -k "file_api or c_driver or reference_test_data or handmade" > gpurun_out/r03k_tests.log 2>&1; tail -3 gpurun_out/r03k_tests.log
timeout -k 10 500 python tools/file_api_times.py --contigs 3000000 > gpurun_out/r03k_file_api_3M.json 2> gpurun_out/r03k.err
tail -3 gpurun_out/r03k.err
