# A/B of two builds of the library inside one gpurun call:
#   bash tools/ab.sh <tag> [bench args...]   (old build: gpurun_in/lib_old.so)
set -x
tag=$1; shift
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary $*"
for r in a b; do
  $B > gpurun_out/${tag}_new_$r.json 2>> gpurun_out/${tag}.err
  GTS_ENGINE_LIB=$PWD/gpurun_in/lib_old.so $B > gpurun_out/${tag}_old_$r.json 2>> gpurun_out/${tag}.err
done
tail -2 gpurun_out/${tag}.err
