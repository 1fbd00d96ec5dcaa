# A/B: expansion kernel with and without the 64-VGPR cap (debug copies), webbase headline K1 time + cage15 share
cd $GRAFT_REPO_ROOT
for v in cap nocap; do
  rm -rf /tmp/dbgroot && mkdir -p /tmp/dbgroot && cp -r pem-spgemm_amd include /tmp/dbgroot/
  if [ $v = nocap ]; then sed -i 's/__launch_bounds__(64 \* S1_XW, 8) s1_expand_kernel/__launch_bounds__(64 * S1_XW) s1_expand_kernel/' /tmp/dbgroot/pem-spgemm_amd/csrc/step1.hip; fi
  (cd /tmp/dbgroot/pem-spgemm_amd/csrc && make clean > /dev/null && make -j8 > /tmp/mk.log 2>&1) || { tail -3 /tmp/mk.log; exit 1; }
  echo "== $v"
  PEM_PKG_ROOT=/tmp/dbgroot python tools/dev/k1_time.py webbase-1M
  PEM_PKG_ROOT=/tmp/dbgroot python tools/dev/k1_time.py cage15 0.125
done
