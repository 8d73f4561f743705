# A/B of the sub-tile count of the narrow filter scatter (profiles/r05_narrow_filter.md); AB_VARIANTS = ablation tags + "default"
for rep in 1 2; do for v in ${AB_VARIANTS:-s111 default}; do
  if [ $v = default ]; then unset AH_LIB_PATH; else export AH_LIB_PATH=$PWD/arrow-rs_amd/lib/ablate/libarrow_hip_$v.so; fi
  echo "== $v rep $rep"; AH_NARROW_ONLY=${AB_CONFIGS:-filter_i32,filter_i16,filter_i8} python bench.py --only-narrow --no-cpu-baseline | python -c "import json,sys; d=json.load(sys.stdin)['configs_narrow']; [print(k, v.get('avg_launch_ms'), v.get('frac'), v.get('error')) for k,v in d.items()]"
done; done
