#!/bin/bash
# stagger / priority experiment on the 4-wave one-launch LoRA kernel
O=gpurun_out/r03b_stagger.txt; : > $O
echo "== trace (default build)" >> $O
TRACE_DUMP=1 AQL_LIB=$PWD/altlib/trace.so python tools/trace_lora.py 2>&1 | grep -v amdgpu.ids >> $O
for v in "" stag1 stag2 prio1 prio3; do
  echo "== variant '$v'" >> $O
  if [ -z "$v" ]; then unset AQL_LIB; else export AQL_LIB=$PWD/altlib/$v.so; fi
  python tools/time_geglu.py 2>&1 | grep -v amdgpu.ids >> $O
  CFGS=auto COLD=1 python tools/tune_lora_cfg.py 2>&1 | grep -v amdgpu.ids >> $O
done
