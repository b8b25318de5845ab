#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 summaries + PMC traffic + SQ counters + phase timers + all configs +
# the kernel-variant A/B. Outputs under gpurun_out/final/ ; scripts/ingest_profiles.py <tag> then copies the summaries into
# profiles/<tag>_*. Every step runs under `timeout` and with stdin closed.
# The profiled command is `bench.py --streams 1` (one capsule, one stream: a launch has the chip to itself and the per-kernel
# averages are what the `roofline` block of the default bench line quotes from its own one-stream leg); the default command
# deals the steps to three streams, where launches of consecutive batches overlap.
set -u
OUT=gpurun_out/final; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T="timeout 300"
P="python bench.py --streams 1 --no-cpu-baseline --no-schedule-legs --no-other-configs"
$T python bench.py --steps 20 --warmup 3 2>/dev/null < /dev/null | grep metric > $OUT/bench.json
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $P --steps 20 --warmup 3 > $OUT/stats_bench.log 2>&1 < /dev/null
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats3 -o s -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-schedule-legs --no-other-configs > /dev/null 2>&1 < /dev/null
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_rep -o s -- $P --same-batch --steps 20 --warmup 3 > /dev/null 2>&1 < /dev/null
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $P --steps 4 --warmup 1 > /dev/null 2>&1 < /dev/null
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $P --steps 4 --warmup 1 > /dev/null 2>&1 < /dev/null
$T rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq1 -o p -- $P --steps 2 --warmup 1 > /dev/null 2>&1 < /dev/null
$T rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_sq2 -o p -- $P --steps 2 --warmup 1 > /dev/null 2>&1 < /dev/null
for c in 3 4 5; do
  $T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c$c -o p -- $P --config $c --steps 2 --warmup 1 > /dev/null 2>&1 < /dev/null
  $T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c$c -o p -- $P --config $c --steps 2 --warmup 1 > /dev/null 2>&1 < /dev/null
done
for k in pipeline fused; do $T python scripts/phase_profile.py 4096 $k 2>&1 < /dev/null | grep -v amdgpu >> $OUT/phase_cycles.txt; done
for c in 3 4 5; do $T python bench.py --config $c --steps 10 --no-other-configs 2>/dev/null < /dev/null | grep metric >> $OUT/configs.jsonl; done
for b in 1 512 4096 16384; do B=$b $T python scripts/pipe_check.py fused pipeline pipeline4 2>&1 < /dev/null | grep -E "^fused|^pipeline" | sed "s/^/batch $b: /" >> $OUT/kernel_variants.txt; done
for S in 1 2 3 4 5 6 8; do $T python bench.py --steps 20 --warmup 3 --streams $S --no-cpu-baseline --no-schedule-legs --no-other-configs 2>/dev/null < /dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams',d['config']['streams'],'value %.3f M solves/s'%(d['value']/1e6),'ms/step %.3f'%d['ms_per_step'])" >> $OUT/streams.txt; done
$T python scripts/snmpc_bench.py 2>&1 < /dev/null | grep -v amdgpu.ids > $OUT/snmpc_bench.txt
for v in prologue-mfma prologue-passes; do echo "== set_kernel(\"$v\") on every capsule (default at ten samples: prologue-mfma)" >> $OUT/snmpc_prologue_variants.txt; SN_PROLOGUE=$v $T python scripts/snmpc_bench.py 2>&1 < /dev/null | grep "^coupled" >> $OUT/snmpc_prologue_variants.txt; done
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sn38 -o s -- python scripts/dev/sn_uph_profile.py 38 38 > /dev/null 2>&1 < /dev/null
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sn05 -o s -- python scripts/dev/sn_uph_profile.py 38 5 > /dev/null 2>&1 < /dev/null
$T python scripts/pcie_inclusive.py 1 3 2>&1 < /dev/null | grep -v amdgpu.ids > $OUT/pcie.txt
$T python scripts/closed_loop_variants.py 2>&1 < /dev/null | grep -v amdgpu.ids > $OUT/closed_loops.txt
# the latency path: wide linearisation / condensing kernels against the ones they replace for small batches; per-kernel times of the 26-vehicle loop
$T python scripts/small_batch_variants.py 2>&1 < /dev/null | grep -v amdgpu.ids > $OUT/small_batch_variants.txt
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/loop26 -o s -- python scripts/loop_prof.py 26 > /dev/null 2>&1 < /dev/null
# the reference's own call pattern: one vehicle, set_initial_state + solve of the mirrored controller class every control step (host wall time)
$T python scripts/probes/controller_step_time.py 2>&1 < /dev/null | grep -v amdgpu.ids > $OUT/controller_step.txt
# round 5: a saved build of an earlier commit (scripts/dev/build_exp_lib.sh <commit>) against the shipped library, interleaved (kernel-level A/B)
for L in exp_libs/lib_*.so; do [ -f "$L" ] && $T python scripts/dev/ab2.py $L shipped 2>&1 < /dev/null | grep -v amdgpu.ids >> $OUT/ab_saved_builds.txt; done
# the rooted gather with the whole iterate (world size 1 under torch.distributed.run: RCCL initialised, the 53 MB slab gathered inside the timed region)
for G in "" "--gather-iterate"; do $T python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --config 4 --scaling strong --global-batch 16384 --steps 10 --warmup 2 --no-cpu-baseline --no-schedule-legs $G 2>/dev/null < /dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config 4, world size 1 under torchrun (nccl = RCCL), gather_iterate', d['config']['gather_iterate'], ': gather %.0f B per rank, value %.3f M solves/s, ms/step %.3f (solve %.3f, gather %.3f)' % (d['config']['gather_bytes_per_rank'], d['value']/1e6, d['ms_per_step'], d['solve_ms_per_step'], d['gather_ms_per_step']))" >> $OUT/gather_iterate.txt; done
for f in 0 1; do echo "TUM_FUSED_EXPAND=$f" >> $OUT/fused_expand.txt; TUM_FUSED_EXPAND=$f $T python scripts/probes/solve_wall_time.py 2>&1 < /dev/null | grep pipeline >> $OUT/fused_expand.txt; TUM_FUSED_EXPAND=$f $T python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-schedule-legs --no-host-legs --no-other-configs 2>/dev/null < /dev/null | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  config 2, default streams: value %.3f M solves/s'%(d['value']/1e6))" >> $OUT/fused_expand.txt; done
# round 6: horizons beyond 40 (six / seven tiles) and a full W against the diagonal one, 4096 instances each
$T python scripts/dev/n56_time.py 2>&1 < /dev/null | grep "^N " > $OUT/long_horizons.txt
$T python scripts/dev/fullw_time.py 2>&1 < /dev/null | grep "W 4096" > $OUT/full_w.txt
$T python scripts/dev/sn_samples_time.py 2>&1 < /dev/null | grep "^samples" > $OUT/snmpc_sample_counts.txt
$T python scripts/dev/ipm4_prof.py 4096 2>&1 < /dev/null | grep -v amdgpu.ids > $OUT/ipm4_phases.txt
head -8 $OUT/stats/s_kernel_stats.csv | cut -c1-150; cut -c1-300 $OUT/bench.json
