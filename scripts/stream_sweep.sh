#!/bin/bash
cd "$(dirname "$0")/.."
P=build/stream_probe
for shape in "12288 4096" "4096 4096" "22016 4096" "4096 11008"; do
  set -- $shape
  for G in 256 512 1024; do timeout 60 $P 0 $1 $2 $G 1; done
  for S in 1 2 4 8; do for D in 1 2 3; do timeout 60 $P 1 $1 $2 $S $D; done; done
  for S in 1 2 4 8; do for D in 1 2 3; do timeout 60 $P 2 $1 $2 $S $D; done; done
  for G in 256 512 1024; do for D in 1 2 3; do timeout 60 $P 3 $1 $2 $G $D; done; done
done
