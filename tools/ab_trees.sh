#!/bin/bash
# A/B of a tool between this tree and an older tree checked out (and built) under .r05tree/: bash tools/ab_trees.sh "<python args>" "<grep pattern>"
ROOT=$PWD
for rep in 1 2; do
  (cd $ROOT && timeout 400 python $1 2>/dev/null | grep "$2" | sed 's/^/this tree | /')
  (cd $ROOT/.r05tree && timeout 400 python $1 2>/dev/null | grep "$2" | sed 's/^/old tree  | /')
done
