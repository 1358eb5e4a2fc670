#!/bin/bash
# tuning builds of the f64 GEMM:  bash tools/gemm_pad_variants.sh db1 "-DRSQP_GEMM_DB=1"  gk32 "-DRSQP_GEMM_GK=32"  pad8 "-DRSQP_GPAD=8"   (CPU side, hipcc only)
# -> restartsqp_amd/lib/librsqp_gemm_<name>.so ; on the GPU box:  RSQP_LIB=restartsqp_amd/lib/librsqp_gemm_db1.so python3 tools/dense_bench.py
cd "$(dirname "$0")/.." || exit 1
O=restartsqp_amd/lib/obj
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-result $defs -x hip -c restartsqp_amd/csrc/dense_la.hip -o /tmp/dense_la_$name.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o restartsqp_amd/lib/librsqp_gemm_$name.so $O/rsqp_api.o $O/qp_small.o $O/qp_large.o $O/sparse.o /tmp/dense_la_$name.o $O/qp_dump.o || exit 1
  echo "built restartsqp_amd/lib/librsqp_gemm_$name.so"
done
