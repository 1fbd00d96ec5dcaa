#!/bin/bash
# Debug copy of the package under /tmp/dbgroot built with -DPEM_S1_DEBUG (GPU box); use with PEM_PKG_ROOT=/tmp/dbgroot
rm -rf /tmp/dbgroot && mkdir -p /tmp/dbgroot && cp -r pem-spgemm_amd include /tmp/dbgroot/ && cd /tmp/dbgroot/pem-spgemm_amd/csrc && make clean > /dev/null && make -j8 EXTRA=-DPEM_S1_DEBUG > /tmp/mk.log 2>&1 || { tail -5 /tmp/mk.log; exit 1; }
