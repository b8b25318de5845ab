#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 summaries + PMC traffic + phase timers + all configs.
# Outputs under gpurun_out/final/ ; scripts/ingest_profiles.py then copies the summaries into profiles/.
set -u
OUT=gpurun_out/final; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 3 2>&1 | grep metric > $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/stats_bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq1 -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_sq2 -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --compare-schedules 2>&1 | grep metric > $OUT/schedules.json
python scripts/phase_profile.py 2>&1 | grep -v amdgpu > $OUT/phase_cycles.txt
python scripts/run_configs.py 2>&1 | grep configs > $OUT/configs.txt
python scripts/pcie_inclusive.py 2>&1 | grep PCIe >> $OUT/configs.txt
python scripts/snmpc_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/snmpc_bench.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sn_stats -o s -- python scripts/snmpc_bench.py > /dev/null 2>&1
head -3 $OUT/stats/s_kernel_stats.csv; cut -c1-200 $OUT/bench.json
