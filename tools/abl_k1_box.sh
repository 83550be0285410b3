# dev tool (profiling build): forward call at sigma 10 / 12.5 / 15 / 20 against the LDS of a K1z workgroup (two box copies)
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
for kb in 31 40 52 64; do for s in 10 12.5 15 20; do echo -n "ZKB=$kb "; EDHIP_ZKB=$kb python tools/time_fwd.py $s 2>&1 | grep -v amdgpu; done; done
