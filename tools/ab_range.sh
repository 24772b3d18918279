#!/bin/bash
# A/B of two library builds on the range proof at DEGREE 22 / LOOKUP_BITS 16 (C++ host), interleaved; see tools/ab_lib.sh
cd $GRAFT_REPO_ROOT
make -C examples -s
L=halo2-scaffold_amd/libh2mi.so
cp $L /tmp/new.so; cp $L.prev /tmp/old.so
for r in 1 2; do
  for v in old new; do
    cp /tmp/$v.so $L
    echo "== $v"
    H2MI_PROOFS=5 ./examples/halo2_lib range 22 16 77 0x5ec2e7 1 | grep steady
    H2MI_PROOFS=8 ./examples/halo2_lib poseidon 20 0 5 0x5ec2e7 1 | grep steady
  done
done
cp /tmp/new.so $L
