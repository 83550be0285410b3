#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05f; rm -rf $O; mkdir -p $O; cd $R; export PYTHONPATH=$R
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/tests.txt 2>&1; tail -30 $O/tests.txt
python bench.py --workload cfg4 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; tail -c 1500 $O/bench_cfg4.json
cp elasticdeform_amd/libedhip.so /tmp/ship.so; cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
timeout 300 python tests/fuzz/fuzz_round4.py 77 60 > $O/fuzz4_exp.txt 2>&1; tail -3 $O/fuzz4_exp.txt
cp /tmp/ship.so elasticdeform_amd/libedhip.so
