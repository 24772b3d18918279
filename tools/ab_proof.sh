#!/bin/bash
# A/B of two builds of libh2mi.so on whole proofs (both hosts), interleaved on one box; see tools/ab_lib.sh for the set-up.
cd $GRAFT_REPO_ROOT
make -C examples -s
L=halo2-scaffold_amd/libh2mi.so
cp $L /tmp/new.so; cp $L.prev /tmp/old.so
for r in 1 2 3; do
  for v in old new; do
    cp /tmp/$v.so $L
    echo "== $v"
    python3 tools/proof_loop.py 20 12 2>/dev/null | tail -1 | cut -c1-90
    H2MI_PROOFS=12 ./examples/standard_plonk 20 0x5ec2e7 5 1 | grep steady
    ${EXTRA:-true}
  done
done
cp /tmp/new.so $L
