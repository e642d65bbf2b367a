# The round's evidence run on ONE box: full GPU suite, default bench (the committed sample), step-only and full-model rocprofv3 kernel
# statistics of the same commands.  Everything lands under gpurun_out/$1.
set -o pipefail
D=gpurun_out/$1
mkdir -p $D && export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $D/pytest.log 2>&1; echo "pytest rc=$?" | tee $D/pytest.rc; tail -3 $D/pytest.log
cp gpurun_out/parity_report.json $D/parity_report.json 2>/dev/null
python bench.py > $D/bench.json 2> $D/bench.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats -d $D/step -o step -- python3 bench.py --no-cpu-baseline --no-cfg5 --no-mixed --no-full-model --no-sustained --steps 100 > $D/bench_step_profiled.json 2> $D/step.err; echo "step prof rc=$?"
python tools/rocpd_stats.py $D/step/step_results.db --skip 10 > $D/step_kernel_stats.csv 2>> $D/step.err
rocprofv3 --kernel-trace --stats -d $D/full -o full -- python3 tools/time_full_model.py > $D/full_model.txt 2> $D/full.err; echo "full prof rc=$?"
python tools/rocpd_stats.py $D/full/full_results.db --skip 3 > $D/full_model_kernel_stats.csv 2>> $D/full.err
rm -rf $D/step $D/full
head -8 $D/step_kernel_stats.csv; head -c 700 $D/bench.json
