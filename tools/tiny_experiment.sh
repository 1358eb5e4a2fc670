#!/bin/bash
# Quick-turnaround tuning build of the hs071-scale tableau kernel (qp_tiny.hip, ~15 s) into restartsqp_amd/lib/librsqp_exp.so;
# select it with RSQP_LIB=<path>. Extra compiler flags as arguments (e.g. -DRSQP_STAMPS).
set -e
cd "$(dirname "$0")/.."
OBJ=restartsqp_amd/lib/obj
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -x hip "$@" -c restartsqp_amd/csrc/qp_tiny.hip -o $OBJ/qp_tiny_exp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o restartsqp_amd/lib/librsqp_exp.so \
    $OBJ/qp_tiny_exp.o $OBJ/qp_lane.o $OBJ/qp_small.o $OBJ/rsqp_api.o $OBJ/qp_large.o $OBJ/sparse.o $OBJ/dense_la.o $OBJ/qp_dump.o $OBJ/rsqp_rccl.o $OBJ/build_stamp.o
echo built restartsqp_amd/lib/librsqp_exp.so
