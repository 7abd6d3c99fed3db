cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for c in 0 1 0 1; do
  export X6LIB=$GRAFT_REPO_ROOT/scripts/_cut$c/libvocr.so
  rm -rf /tmp/x6c; rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/x6c -- python3 scripts/x6_only.py > /dev/null 2>&1
  echo "== variant $c"; python scripts/x6_clock.py /tmp/x6c
done
