# A/B of library builds on ONE box, alternating: AB_VARIANTS = ablation tags (arrow-rs_amd/lib/ablate/libarrow_hip_<tag>.so) + "default";
# AB_CMD = the bench command; AB_PICK = a python expression over the parsed JSON line `d` to print
for rep in 1 2; do for v in ${AB_VARIANTS:-default}; do
  if [ $v = default ]; then unset AH_LIB_PATH; else export AH_LIB_PATH=$PWD/arrow-rs_amd/lib/ablate/libarrow_hip_$v.so; fi
  echo "== $v rep $rep"; ${AB_CMD} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(${AB_PICK})"
done; done
