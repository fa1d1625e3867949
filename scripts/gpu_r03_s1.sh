#!/bin/bash
# round 3, GPU session 1: new -m gpu tests, the default bench line (with `extra`), the --gpus 2 refusal on a 1-GPU box,
# the strip emission ablation (plain stores instead of atomics), a kernel trace of the headline call
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/s1; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "kron26 or scale16_vs_scipy or bench_sizes or (rmat_vs_oracle and 20)" ) > $O/tests_new.log 2>&1
tail -5 $O/tests_new.log
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.err
python bench.py --gpus 2 > $O/gpus2.out 2> $O/gpus2.err; echo "gpus2 exit=$?" >> $O/gpus2.err; cat $O/gpus2.err
bash scripts/variants_ab.sh run > $O/strip_emission_variants.txt 2>&1; cat $O/strip_emission_variants.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_s1 -o headline -- python $OLDPWD/bench.py --no-extra --no-cpu-baseline > $OLDPWD/$O/bench_under_rocprof.json 2>/dev/null )
find /tmp/prof_s1 -name "*kernel_stats.csv" -exec cp {} $O/headline_kernel_stats.csv \;
head -8 $O/headline_kernel_stats.csv
