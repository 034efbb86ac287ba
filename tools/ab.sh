#!/bin/bash
# A/B of one variant library against the product library in the same GPU run (boxes differ by a few percent):
#   tools/ab.sh VARIANT [kbench filter]     -> alternates scmomer_amd/libscmomer_hip_VARIANT.so and the product library
V=$1; F=$2
for i in 1 2; do
  SCMOMER_HIP_LIB=$PWD/scmomer_amd/libscmomer_hip_$V.so timeout -k 10 300 python tools/kbench.py $F | sed "s/^/$V /" || exit 1
  timeout -k 10 300 python tools/kbench.py $F | sed 's/^/product /' || exit 1
done
