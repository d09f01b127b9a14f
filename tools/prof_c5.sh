# C5 (DPD, V = 10 000, K = 8192): the bench line, one sub-sweep's launches in
# order, and three counter passes (SQ issue / LDS + VMEM / scalar + misc) of
# k_vs_stream; on the GPU box:  bash tools/prof_c5.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?not on a gpurun box}"
B="bench.py --cpu-rows 0 --other-batches= --other-configs= --no-breakdown --steps 3 --warmup 2 --config dpd --groups 8192 --dim 10000"
python3 $B 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value']/1e9, d['ms_per_step'], d['roofline'].get('avg_launch_ms'))"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/t_c5 -- python3 $B > /dev/null 2>&1
python3 tools/batch_timeline.py gpurun_out/t_c5 k_vs_stream
rm -rf gpurun_out/t_c5
for ctr in "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAVES"; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/p_c5 -- python3 $B > /dev/null 2> gpurun_out/p.log
  python3 tools/pmc_summary.py gpurun_out/p_c5 k_vs_stream
  rm -rf gpurun_out/p_c5
done
