# dev tool: forward / gradient call at sigma 5 .. 20 with the shipped library (the policies as shipped)
for s in 5 10 12.5 15 20; do python tools/time_fwd.py $s 2>&1 | grep -v amdgpu; python tools/time_grad.py $s 2>&1 | grep -v amdgpu; done
