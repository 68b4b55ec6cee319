#!/bin/bash
# Counter evidence for the generator kernels: one rocprofv3 --pmc pass per counter group over
# one bench step, then tools/pmc_g16_table.py sums every counter per kernel template.
# usage (through gpurun): tools/run_pmc_g16.sh <tag> [bench.py arguments, e.g. --workload C2 --batch 1]
set -eu
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; TAG="${1:-pmc_g16}"; shift || true
GROUPS_=(
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU"
  "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_F16"
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum GRBM_GUI_ACTIVE"
  "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES"
)
i=0
for G in "${GROUPS_[@]}"; do
  O="$R/gpurun_out/$TAG/g$i"; rm -rf "$O"; mkdir -p "$O"
  rocprofv3 --pmc $G --output-format csv -d "$O" -o p -- python3 "$R/bench.py" --in-flight 1 "$@" --steps 1 --warmup 1 --profile-steps 0 --no-cpu-baseline > /dev/null 2> "$O/err.txt" || echo "pass $i failed: $(tail -2 $O/err.txt)"
  i=$((i+1))
done
python3 "$R/tools/pmc_g16_table.py" $(ls "$R"/gpurun_out/$TAG/g*/p_counter_collection.csv) > "$R/gpurun_out/$TAG/table.txt" || true
rm -f "$R"/gpurun_out/$TAG/g*/p_agent_info.csv
cat "$R/gpurun_out/$TAG/table.txt"
