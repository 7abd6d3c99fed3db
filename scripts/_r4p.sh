cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -w -std=c++17 -I include -I vistaocr_amd/csrc -o /tmp/lstm_stamp4 scripts/lstm_stamp4.hip && /tmp/lstm_stamp4 32
