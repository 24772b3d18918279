#!/bin/bash
# A/B of two library builds on the C++ host's proofs at 2^8 / 2^16 / 2^20, interleaved on one box (see tools/ab_lib.sh)
cd $GRAFT_REPO_ROOT
make -C examples -s
L=halo2-scaffold_amd/libh2mi.so
cp $L /tmp/new.so; cp $L.prev /tmp/old.so
for r in 1 2 3; do
  for v in old new; do
    cp /tmp/$v.so $L
    echo -n "== $v "
    for k in 8 16 20; do H2MI_PROOFS=30 ./examples/standard_plonk $k 0x5ec2e7 5 1 | grep steady | awk '{printf "k%s %s  ", "'$k'", $2}'; done; echo
  done
done
cp /tmp/new.so $L
