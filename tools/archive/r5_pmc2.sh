# round 5: cache traffic of the three inter kernels: L1 accesses, L1 -> L2 requests, L2 hits / misses, HBM requests
R=$GRAFT_REPO_ROOT; T=${1:-r5pmc2}
one() { bash $R/tools/prof_pmc.sh ${T}_$1 cfg4_main_8k_10b_ra "$2" > /dev/null 2>&1; grep -E "k_inter|k_addb_alf" $R/gpurun_out/${T}_$1.csv | cut -c1-120; }
one f "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
one g "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_HIT_sum TCC_MISS_sum"
one h "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
one i "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
