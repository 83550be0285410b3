# dev tool (profiling build): gradient call at sigma 10 / 12.5 / 15 / 20 against the size of K2's cell block
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
for kb in 36 48 56 64; do for s in 10 12.5 15 20; do echo -n "GRAD_BOX_KB=$kb "; EDHIP_GRAD_BOX_KB=$kb python tools/time_grad.py $s 2>&1 | grep -v amdgpu; done; done
