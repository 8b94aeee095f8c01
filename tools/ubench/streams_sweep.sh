#!/bin/bash
cd "$(dirname "$0")"
./streams sweep 512
./streams sweep 256
for pad in 0 37 517 2049 1 100; do ./streams 512 $pad 0 2>&1 | grep -o "sep: .*\|\"sep\": [0-9.]*" | head -2 | tr '\n' ' '; echo; done
