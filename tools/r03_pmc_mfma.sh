#!/bin/bash
# MFMA pipe utilisation (rocprofv3 counters) of the attention kernels and the LoRA / conv dominant kernels
for w in "attn 4 4096 8" "attn 8 4096 8" "attnq 4 4096 8" "attnq 8 4096 8" "geglu 32768 1280 320" "lora 32768 320 320" "conv 8 64 320 320" "chain 32768"; do
  tag=$(echo $w | tr ' ' '_')
  PMC_GROUP="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE" tools/pmc_kernels.sh mfma_$tag tools/pmc_one.py $w > /dev/null 2>&1
  echo "== pmc_one.py $w"
  python tools/pmc_mfma_util.py /tmp/pmck_mfma_$tag
done
