#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats and the two PMC passes (FETCH_SIZE / WRITE_SIZE need separate
# passes: TCC slots) of the same bench command.  Outputs under gpurun_out/prof/<tag>_{stats,fetch,write}/.
#   usage: tools/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ARGS=${*:---steps 8 --warmup 2 --no-cpu-baseline --easy-noise 0 --min-seconds 0 --no-single-stream}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
python -c "import sys; sys.path.insert(0, '$REPO'); import bench; print(bench.csrc_sha16())" > $OUT/${TAG}_csrc_sha16.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python $REPO/bench.py $ARGS > $OUT/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_fetch -- python $REPO/bench.py $ARGS > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_write -- python $REPO/bench.py $ARGS > $OUT/${TAG}_write.log 2>&1
# VALU issue / lane utilisation / occupancy (SQ block: own passes)
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_valu -- python $REPO/bench.py $ARGS > $OUT/${TAG}_valu.log 2>&1
rocprofv3 --pmc VALUUtilization VALUBusy MeanOccupancyPerCU --kernel-trace --output-format csv -d $OUT/${TAG}_util -- python $REPO/bench.py $ARGS > $OUT/${TAG}_util.log 2>&1
# fp64 flop count (SQ instruction counters by type; lanes active from SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU) and MFMA use
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/${TAG}_fp64 -- python $REPO/bench.py $ARGS > $OUT/${TAG}_fp64.log 2>&1
# where a wave's cycles go: issue vs wait
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/${TAG}_wait -- python $REPO/bench.py $ARGS > $OUT/${TAG}_wait.log 2>&1
find $OUT -name "*.csv" | head -40
tail -1 $OUT/${TAG}_stats.log | cut -c1-300
