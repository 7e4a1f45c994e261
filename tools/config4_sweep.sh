mkdir -p gpurun_out/r04
for cfg in "SPREAD=-1" "SPREAD=0" "SPREAD=0 PPL=1" "SPREAD=-1 GPU_MAX_HW_QUEUES=8"; do
  echo "== $cfg"
  env $cfg ROOTS=256,32 GROUPS=1,2,4,8 OUT=/dev/null timeout -k 10 300 python3 tools/config4_pipeline.py 6 2>&1 | grep roots | cut -c1-110
done
