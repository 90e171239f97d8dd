#!/bin/bash
S1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"
S2="SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE"
bash scripts/gpu_pmc.sh c3 "RqsF<float, false>" "$S1" "$S2"
bash scripts/gpu_pmc.sh c3 "RqsF<float, true>" "$S1" "$S2"
bash scripts/gpu_pmc.sh c5a seq_kernel "$S1" "$S2"
bash scripts/gpu_pmc.sh c5b chol_inv "$S1" "$S2"
bash scripts/gpu_pmc.sh c2 chain_flat "$S1" "$S2"
