cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
run() { echo "== $*"; env "$@" python tools/time_fwd.py; }
run A=0
run EDHIP_ZNGEN=2048
run EDHIP_ZNGEN=640
run EDHIP_ZNGEN=320
run EDHIP_ZKB=31
run EDHIP_ZKB=31 EDHIP_ZNGEN=640
