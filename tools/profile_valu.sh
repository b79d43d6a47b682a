TAG=r01d
REPO=$(pwd); OUT=$REPO/gpurun_out/prof; mkdir -p $OUT
ARGS="--steps 8 --warmup 2 --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_valu -- python $REPO/bench.py $ARGS > $OUT/${TAG}_valu.log 2>&1
rocprofv3 --pmc VALUUtilization VALUBusy MeanOccupancyPerCU --kernel-trace --output-format csv -d $OUT/${TAG}_util -- python $REPO/bench.py $ARGS > $OUT/${TAG}_util.log 2>&1
tail -2 $OUT/${TAG}_valu.log | cut -c1-200; tail -2 $OUT/${TAG}_util.log | cut -c1-200
