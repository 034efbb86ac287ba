#!/bin/bash
# Builds a variant of the library next to the product one: tools/build_variant.sh NAME "-DFLAG ..."  ->
# scmomer_amd/libscmomer_hip_NAME.so (select it with SCMOMER_HIP_LIB=...); objects under csrc/build_NAME.
NAME=$1; DEFS=$2
cd "$(dirname "$0")/../scmomer_amd/csrc" && mkdir -p build_$NAME
for f in gemm ffn_chain norm_embed favor favor_free s1_tail ingest optim cell_tail distill_tail capi; do
  EXTRA=""; [ $f = favor -o $f = gemm ] && EXTRA="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $EXTRA $DEFS -c $f.hip -o build_$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_$NAME/*.o -o ../libscmomer_hip_$NAME.so && echo built libscmomer_hip_$NAME.so
