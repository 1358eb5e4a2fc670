#!/bin/bash
# Quick-turnaround tuning build of the 8-lane LDS kernel only (-DRSQP_SMALL_EXPERIMENT: ~20 s instead of
# ~2 min for all instantiations) into restartsqp_amd/lib/librsqp_exp.so; select it with RSQP_LIB=<path>.
# EXPDEF=-DRSQP_SMALL_EXPERIMENT=2 builds only the four-wave explicit-inverse kernel (EngineX<256>).
set -e
cd "$(dirname "$0")/.."
OBJ=restartsqp_amd/lib/obj
mkdir -p $OBJ
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -x hip ${EXPDEF:--DRSQP_SMALL_EXPERIMENT=1} "$@" \
    -c restartsqp_amd/csrc/qp_small.hip -o $OBJ/qp_small_exp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o restartsqp_amd/lib/librsqp_exp.so \
    $OBJ/qp_small_exp.o $OBJ/qp_tiny.o $OBJ/qp_lane.o $OBJ/rsqp_api.o $OBJ/qp_large.o $OBJ/sparse.o $OBJ/dense_la.o $OBJ/qp_dump.o $OBJ/rsqp_rccl.o $OBJ/build_stamp.o
echo built restartsqp_amd/lib/librsqp_exp.so
