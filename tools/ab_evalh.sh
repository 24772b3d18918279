cd $GRAFT_REPO_ROOT
L=halo2-scaffold_amd/libh2mi.so
cp $L /tmp/new.so; cp $L.prev /tmp/old.so
for v in old new old new; do cp /tmp/$v.so $L; echo "== $v"; python3 tools/poly_sweep.py 20 2>/dev/null | grep "evaluate_h" | cut -c1-140; done
cp /tmp/new.so $L
