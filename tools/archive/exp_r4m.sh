# which unit of the memory pipeline is busiest in k_inter / k_addb_alf: texture addresser (TA), texture data (TD), vector L1 (TCP) stall reasons, L2 busy
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ctrs in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TD_TD_BUSY_sum TA_BUSY_avr" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TD_TC_STALL_sum" \
            "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
            "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_TCP_LATENCY_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TD_LOAD_WAVEFRONT_sum" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR" "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  rm -rf $R/gpurun_out/pmc_x
  EXP_STEPS=3 timeout -k 5 200 rocprofv3 --pmc $ctrs -d $R/gpurun_out/pmc_x -o p -- python $R/tools/exp_inter_order.py 16:0 > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc_x -name "*.db" | head -1) /dev/null | grep -E "k_inter|k_addb_alf" | sed 's/(AlfArgs.*)"/"/; s/(InterArgs)//' | cut -c1-100
done
rm -rf $R/gpurun_out/pmc_x
