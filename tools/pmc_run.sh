#!/bin/bash
# PMC passes over tools/pmc_one.py (one counter group per pass, kernel-trace only).  usage: tools/pmc_run.sh <tag> <pmc_one args...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
G2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"
G3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS"
G4="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TCP_TCC_READ_REQ_LATENCY_sum"
G5="FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_TAG_STALL_sum"
i=0
for G in "$G1" "$G2" "$G3" "$G4" "$G5"; do
  i=$((i+1)); rm -rf /tmp/pmc_${tag}_$i
  if [ -n "$PMC_ONLY" ] && [[ " $PMC_ONLY " != *" $i "* ]]; then continue; fi   # e.g. PMC_ONLY="1 3": only those groups
  timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d /tmp/pmc_${tag}_$i -o run -- python $GRAFT_REPO_ROOT/tools/pmc_one.py "$@" > /tmp/pmc_${tag}_$i.log 2>&1
done
python $GRAFT_REPO_ROOT/tools/pmc_parse.py /tmp/pmc_${tag}_ 5 > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}.txt
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}.txt
