set -x
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
$B > gpurun_out/r03w_64s3.json 2> gpurun_out/r03w.err
$B --opt batch_big_slots=4 > gpurun_out/r03w_64s4.json 2>> gpurun_out/r03w.err
$B --opt batch_big_contigs=32 > gpurun_out/r03w_32s3.json 2>> gpurun_out/r03w.err
$B > gpurun_out/r03w_64s3b.json 2>> gpurun_out/r03w.err
$B --opt batch_big_slots=4 > gpurun_out/r03w_64s4b.json 2>> gpurun_out/r03w.err
$B --opt batch_big_contigs=32 > gpurun_out/r03w_32s3b.json 2>> gpurun_out/r03w.err
$B --opt pool_fill_kb=6 > gpurun_out/r03w_fill6.json 2>> gpurun_out/r03w.err
tail -2 gpurun_out/r03w.err
