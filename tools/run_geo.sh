# dev tool: geometry kernel phases (profiling build) + forward call timing (shipped build) + K1 parity subset
cp elasticdeform_amd/libedhip.so /tmp/ship.so; cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
python tools/geo_phases.py 5 2>&1 | grep -v amdgpu
cp /tmp/ship.so elasticdeform_amd/libedhip.so
python tools/time_fwd.py 5 2>&1 | grep -v amdgpu
export TMPDIR=/tmp; cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/pg -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/time_fwd.py 5 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; python tools/kernel_stats_csv.py /tmp/pg/p_kernel_stats.csv | cut -c1-140 | head -8
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or k1 or raw or RAW or grid" 2>&1 | tail -3
