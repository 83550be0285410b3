# dev tool: K1z tile classes at sigma 7.5 .. 15 (counters build) with the standard and the large boxes; timing of both for forward and gradient
cp elasticdeform_amd/libedhip.so /tmp/ship.so
cp tools/libedhip_stats.so elasticdeform_amd/libedhip.so
for lb in 0 1; do for s in 7.5 10 12.5 15; do echo -n "LARGE_BOXES=$lb "; EDHIP_LARGE_BOXES=$lb python tools/k1_stats.py 256 3 $s 2>&1 | grep k1z; done; done
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
for lb in 0 1; do for s in 5 7.5 10; do echo -n "LARGE_BOXES=$lb "; EDHIP_LARGE_BOXES=$lb python tools/time_fwd.py $s 2>&1 | grep -v amdgpu;  echo -n "LARGE_BOXES=$lb "; EDHIP_LARGE_BOXES=$lb python tools/time_grad.py $s 2>&1 | grep -v amdgpu; done; done
cp /tmp/ship.so elasticdeform_amd/libedhip.so
