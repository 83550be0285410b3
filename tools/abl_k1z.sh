cp tools/libedhip_exp.so elasticdeform_amd/libedhip.so
run() { echo "== $*"; env "$@" KSTATS_ROWS=8 tools/kstats.sh x tools/prof_fwd.py 5 | grep "k1z" | cut -c1-100; }
run A=1
run EDHIP_TILE_DBG=8388608
run EDHIP_TILE_DBG=2130706432
