# A/B of engine options inside one gpurun call:
#   bash tools/opt_ab.sh <tag> name=value[,name=value...] [...]   (first run of each round: defaults)
# GTS_AB_ARGS: extra bench.py arguments for every run
set -x
tag=$1; shift
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary $GTS_AB_ARGS"
for r in a b; do
  $B > gpurun_out/${tag}_default_$r.json 2>> gpurun_out/${tag}.err
  for o in "$@"; do
    args=""
    for kv in ${o//,/ }; do args="$args --opt $kv"; done
    name=${o//=/_}; name=${name//,/+}
    $B $args > gpurun_out/${tag}_${name}_$r.json 2>> gpurun_out/${tag}.err
  done
done
tail -2 gpurun_out/${tag}.err
