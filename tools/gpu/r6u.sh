cd $GRAFT_REPO_ROOT
run() { echo "--- $1"; env $1 M3D_C5_REPS=4 python tools/time_c5_plain.py 2>&1 | tail -2 | cut -c1-260; }
run "M3D_X=0"
run "M3D_DBG_NO_OWN=1"
run "M3D_DBG_NO_OWN=1 M3D_GPB=16 M3D_SCORE_MIN_WGS=1024"
run "M3D_DBG_NO_OWN=1 M3D_GPB=16 M3D_SCORE_MIN_WGS=4096"
run "M3D_DBG_NO_OWN=1 M3D_GPB=8 M3D_SCORE_MIN_WGS=2048"
