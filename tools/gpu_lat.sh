#!/bin/bash
for o in 0 60 240; do build/cycle_latency 300 $o 2>&1 | grep "^N="; done
