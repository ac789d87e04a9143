# A/B of engine options inside one gpurun call:
#   bash tools/opt_ab.sh <tag> name=value [name=value ...]   (first run of each pair: defaults)
set -x
tag=$1; shift
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
for r in a b; do
  $B > gpurun_out/${tag}_default_$r.json 2>> gpurun_out/${tag}.err
  for o in "$@"; do
    $B --opt $o > gpurun_out/${tag}_${o/=/_}_$r.json 2>> gpurun_out/${tag}.err
  done
done
tail -2 gpurun_out/${tag}.err
