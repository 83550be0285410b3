# dev tool: gradient call of the K2 variants in the profiling build (tools/time_grad.py), sigma 5 / 10 / 15
cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so   # (on the GPU box only: the snapshot there is a throw-away copy)
run() { echo "== $*"; for s in 5 10 15; do env "$@" python tools/time_grad.py $s; done; }
run EDHIP_NONE=1     # hot_grad_kernel
run EDHIP_K2Z=1
run EDHIP_K2Y=1
run EDHIP_K2S=1
