#!/bin/bash
for o in 60 240; do
echo "--- new O=$o"; build/cycle_latency 300 $o 2>&1 | grep "^N="
echo "--- prev O=$o"; LD_LIBRARY_PATH=$PWD/build/prev build/cycle_latency 300 $o 2>&1 | grep "^N="
done
