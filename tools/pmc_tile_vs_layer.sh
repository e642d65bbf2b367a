set -o pipefail
mkdir -p gpurun_out/r06j && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS"
P3="SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d gpurun_out/r06j/p1 -- python3 tools/profile_tile_kernels.py --tile-conv > gpurun_out/r06j/p1.log 2>&1 && \
rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d gpurun_out/r06j/p2 -- python3 tools/profile_tile_kernels.py --tile-conv > gpurun_out/r06j/p2.log 2>&1 && \
rocprofv3 --pmc $P3 --kernel-trace --output-format csv -d gpurun_out/r06j/p3 -- python3 tools/profile_tile_kernels.py --tile-conv > gpurun_out/r06j/p3.log 2>&1
for k in gatv2_layer_conv_kernel gatv2_tile_conv_kernel; do echo "== $k"; python3 tools/pmc_sum.py $k gpurun_out/r06j/p1 gpurun_out/r06j/p2 gpurun_out/r06j/p3; done > gpurun_out/r06j/sq_counters.txt 2>&1
cat gpurun_out/r06j/sq_counters.txt
rm -rf gpurun_out/r06j/p1 gpurun_out/r06j/p2 gpurun_out/r06j/p3
