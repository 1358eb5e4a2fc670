#!/bin/bash
# tuning builds of the f64 GEMM with other LDS row paddings:  bash tools/gemm_pad_variants.sh 8 12 16   (CPU side, hipcc only)
# -> restartsqp_amd/lib/librsqp_gpad<N>.so ; on the GPU box:  RSQP_LIB=restartsqp_amd/lib/librsqp_gpad16.so python3 tools/dense_bench.py
cd "$(dirname "$0")/.." || exit 1
O=restartsqp_amd/lib/obj
for g in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-result -DRSQP_GPAD=$g -x hip -c restartsqp_amd/csrc/dense_la.hip -o /tmp/dense_la_gpad$g.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o restartsqp_amd/lib/librsqp_gpad$g.so $O/rsqp_api.o $O/qp_small.o $O/qp_large.o $O/sparse.o /tmp/dense_la_gpad$g.o $O/qp_dump.o || exit 1
  echo "built restartsqp_amd/lib/librsqp_gpad$g.so"
done
