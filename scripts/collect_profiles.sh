#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 summaries + PMC traffic + SQ counters + phase timers + all configs +
# the kernel-variant A/B. Outputs under gpurun_out/final/ ; scripts/ingest_profiles.py <tag> then copies the summaries into
# profiles/<tag>_*. Every step runs under `timeout` and with stdin closed.
set -u
OUT=gpurun_out/final; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T="timeout 300"
$T python bench.py --steps 20 --warmup 3 2>/dev/null < /dev/null | grep metric > $OUT/bench.json
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-schedule-legs > $OUT/stats_bench.log 2>&1 < /dev/null
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-schedule-legs > /dev/null 2>&1 < /dev/null
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-schedule-legs > /dev/null 2>&1 < /dev/null
$T rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq1 -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-schedule-legs > /dev/null 2>&1 < /dev/null
$T rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_sq2 -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-schedule-legs > /dev/null 2>&1 < /dev/null
for k in pipeline fused; do $T python scripts/phase_profile.py 4096 $k 2>&1 < /dev/null | grep -v amdgpu >> $OUT/phase_cycles.txt; done
for c in 3 4 5; do $T python bench.py --config $c --steps 10 2>/dev/null < /dev/null | grep metric >> $OUT/configs.jsonl; done
for b in 1 512 4096 16384; do B=$b $T python scripts/pipe_check.py fused pipeline 2>&1 < /dev/null | grep -E "^fused|^pipeline" | sed "s/^/batch $b: /" >> $OUT/kernel_variants.txt; done
# the interior point kernel bounded to two wavefronts per SIMD (exp_libs/lib_wps2.so: -DIPM_WPS=2), OCPs per CU swept through the LDS request
if [ -f exp_libs/lib_wps2.so ]; then
  for l in 0 32768 40960; do echo "IPM_WPS=2, dynamic LDS request $l B:" >> $OUT/ipm_occupancy.txt; TUM_NMPC_LIB=$PWD/exp_libs/lib_wps2.so TUM_IPM_LDS=$l $T python scripts/pipe_check.py pipeline 2>&1 < /dev/null | grep -E "^pipeline" >> $OUT/ipm_occupancy.txt; done
  echo "IPM_WPS=1 (shipped):" >> $OUT/ipm_occupancy.txt; $T python scripts/pipe_check.py pipeline 2>&1 < /dev/null | grep -E "^pipeline" >> $OUT/ipm_occupancy.txt
fi
$T python scripts/snmpc_bench.py 2>&1 < /dev/null | grep -v amdgpu.ids > $OUT/snmpc_bench.txt
$T python scripts/pcie_inclusive.py 2>&1 < /dev/null | grep PCIe > $OUT/pcie.txt
head -6 $OUT/stats/s_kernel_stats.csv | cut -c1-150; cut -c1-200 $OUT/bench.json
