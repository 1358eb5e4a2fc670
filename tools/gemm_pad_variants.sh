#!/bin/bash
# tuning builds of the f64 GEMM (LDS double buffering on / off; edit the -D line for other row paddings):  bash tools/gemm_pad_variants.sh 1   (CPU side, hipcc only)
# -> restartsqp_amd/lib/librsqp_gemmdb<N>.so ; on the GPU box:  RSQP_LIB=restartsqp_amd/lib/librsqp_gemmdb1.so python3 tools/dense_bench.py
cd "$(dirname "$0")/.." || exit 1
O=restartsqp_amd/lib/obj
for g in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-result -DRSQP_GPAD=4 -DRSQP_GEMM_DB=$g -x hip -c restartsqp_amd/csrc/dense_la.hip -o /tmp/dense_la_db$g.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o restartsqp_amd/lib/librsqp_gemmdb$g.so $O/rsqp_api.o $O/qp_small.o $O/qp_large.o $O/sparse.o /tmp/dense_la_db$g.o $O/qp_dump.o || exit 1
  echo "built restartsqp_amd/lib/librsqp_gemmdb$g.so"
done
