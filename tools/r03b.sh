set -x
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
$B --opt batch_big_contigs=0 > gpurun_out/r03b_big0.json 2> gpurun_out/r03b.err &&
$B --opt batch_big_contigs=256 > gpurun_out/r03b_big256.json 2>> gpurun_out/r03b.err &&
$B --opt batch_big_contigs=128 --opt batch_big_slots=4 > gpurun_out/r03b_big128s4.json 2>> gpurun_out/r03b.err &&
$B --opt batch_big_contigs=64 --opt batch_big_slots=4 > gpurun_out/r03b_big64s4.json 2>> gpurun_out/r03b.err &&
$B --inversions 0.1 --duplicate-pairs --opt batch_big_contigs=0 > gpurun_out/r03b_inv.json 2>> gpurun_out/r03b.err &&
$B --inversions 0.1 --duplicate-pairs --opt batch_big_contigs=0 --opt defer_ref_min_contigs=0 > gpurun_out/r03b_inv_nolate.json 2>> gpurun_out/r03b.err
python -m pytest tests/test_gpu_parity.py -x -q -k "not full_size and not 50M" > gpurun_out/r03b_tests.log 2>&1; tail -3 gpurun_out/r03b_tests.log
