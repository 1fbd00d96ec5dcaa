#!/bin/bash
# Debug copy of the package under /tmp/dbgroot built with extra flags (GPU box); use with PEM_PKG_ROOT=/tmp/dbgroot
# usage: tools/dbg_build.sh [EXTRA flags, default -DPEM_S1_DEBUG]
FLAGS=${1:--DPEM_S1_DEBUG}
rm -rf /tmp/dbgroot && mkdir -p /tmp/dbgroot && cp -r pem-spgemm_amd include /tmp/dbgroot/ && cd /tmp/dbgroot/pem-spgemm_amd/csrc && make clean > /dev/null && make -j8 EXTRA="$FLAGS" > /tmp/mk.log 2>&1 || { tail -5 /tmp/mk.log; exit 1; }
