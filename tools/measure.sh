# The measurements of a round on its final build (GPU box): the seven rocprofv3 passes of tools/profile.sh + the bench variants.
#   usage: bash tools/measure.sh <tag>      -> gpurun_out/<tag>_*
TAG=${1:-r06}
set -u
mkdir -p gpurun_out
bash tools/profile.sh ${TAG} > gpurun_out/${TAG}_profile.log 2>&1
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2>/dev/null
python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --streams 1 --batch 1024 > gpurun_out/${TAG}_bench_slice1.json 2>/dev/null
python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --batch 32768 > gpurun_out/${TAG}_bench_b32768.json 2>/dev/null
python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --batch 8192 > gpurun_out/${TAG}_bench_b8192.json 2>/dev/null
python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --workload atlas20_freefall > gpurun_out/${TAG}_bench_freefall.json 2>/dev/null
python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --workload atlas33_contact --batch 8192 > gpurun_out/${TAG}_bench_atlas33.json 2>/dev/null
python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --workload atlas33_contact --rollout 64 --batch 8192 > gpurun_out/${TAG}_bench_atlas33_rollout.json 2>/dev/null
python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --workload atlas33_contact --rollout 64 --batch 8192 --graph > gpurun_out/${TAG}_bench_atlas33_rollout_graph.json 2>/dev/null
python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --workload atlas33_contact --rollout 64 --batch 8192 --checkpoint-every 16 > gpurun_out/${TAG}_bench_atlas33_rollout_checkpoint16.json 2>/dev/null
python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --rollout 64 > gpurun_out/${TAG}_bench_atlas20_rollout.json 2>/dev/null
python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --rollout 64 --graph > gpurun_out/${TAG}_bench_atlas20_rollout_graph.json 2>/dev/null
python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --max-contacts 16 > gpurun_out/${TAG}_bench_48rows.json 2>/dev/null
python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 --max-contacts 24 --steps 8 --warmup 2 > gpurun_out/${TAG}_bench_general.json 2>/dev/null
NBL_FUSED_DETECT=0 python bench.py --no-cpu-baseline --no-single-stream --easy-noise 0 > gpurun_out/${TAG}_bench_unfused_detect.json 2>/dev/null
for f in gpurun_out/${TAG}_bench_*.json; do echo $f; tail -1 $f | cut -c1-260; done
