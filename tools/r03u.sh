set -x
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
$B > gpurun_out/r03u_128s4.json 2> gpurun_out/r03u.err
$B --opt batch_big_contigs=0 > gpurun_out/r03u_big0.json 2>> gpurun_out/r03u.err
$B --opt batch_big_contigs=64 --opt batch_big_slots=4 > gpurun_out/r03u_64s4.json 2>> gpurun_out/r03u.err
$B --opt batch_big_contigs=64 --opt batch_big_slots=8 > gpurun_out/r03u_64s8.json 2>> gpurun_out/r03u.err
$B --opt batch_big_contigs=256 --opt batch_big_slots=8 > gpurun_out/r03u_256s8.json 2>> gpurun_out/r03u.err
$B --opt pool_fill_kb=6 > gpurun_out/r03u_fill6.json 2>> gpurun_out/r03u.err
tail -2 gpurun_out/r03u.err
