cd $GRAFT_REPO_ROOT
python scripts/conv_occ.py 2>&1 | grep -v amdgpu
VOCR_CONV_LDS_PAD=40000 python scripts/conv_occ.py 2>&1 | grep -v amdgpu
VOCR_CONV_LDS_PAD=100000 python scripts/conv_occ.py 2>&1 | grep -v amdgpu
