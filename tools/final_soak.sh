# The randomised soaks on the final build of a round (criterion: a world above 1e-6 must be proven reference-unstable; NBL_SOAK_TOL=1e-7: one CFM world at 1.2e-7 is left over) (GPU box): every mode of tools/soak_parity.py and tools/soak_stress.py, the warm-start
# and Jacobian soaks.  Prints one totals line per run.
set -u
S=${1:-0}   # seed offset: a second pass with other seeds is `bash tools/final_soak.sh 100000`
for m in "" big multi balls far; do echo "parity:$m $(python tools/soak_parity.py $((40000+S)) 300 256 $m 2>&1 | tail -1)"; done
echo "warm $(python tools/soak_warm.py $((41000+S)) 300 256 balls 2>&1 | tail -1)"
echo "jacobians $(python tools/soak_jacobians.py $((42000+S)) 100 4 2>&1 | tail -1)"
for mode in dt tinydt fast torque mass nograv geom mu subset atlimit capsule limits selfcol adjacent; do echo "stress:$mode $(python tools/soak_stress.py $mode $((43000+S)) 120 256 2>&1 | tail -1)"; done
for v in balls big multi; do echo "stress:mix:$v $(python tools/soak_stress.py mix $((44000+S)) 200 256 $v 2>&1 | tail -1)"; done
echo "warm:mix $(python tools/soak_warm.py $((45000+S)) 300 256 balls mix 2>&1 | tail -1)"
for v in balls big; do echo "jacobians:mix:$v $(python tools/soak_jacobians.py $((46000+S)) 150 4 $v mix 2>&1 | tail -1)"; done
