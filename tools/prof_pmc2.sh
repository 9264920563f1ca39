R=$GRAFT_REPO_ROOT
bash $R/tools/prof_pmc.sh ta1 cfg4_main_8k_10b_ra "GRBM_GUI_ACTIVE TA_TA_BUSY_sum" | grep -E "k_inter|k_alf|k_addb"
bash $R/tools/prof_pmc.sh ta2 cfg4_main_8k_10b_ra "TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" | grep -E "k_inter|k_alf|k_addb"
bash $R/tools/prof_pmc.sh sq2 cfg4_main_8k_10b_ra "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY" | grep -E "k_inter|k_alf|k_addb"
