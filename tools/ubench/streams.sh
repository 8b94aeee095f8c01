#!/bin/bash
# tools/ubench/streams.hip in fresh processes (physical pages differ from process to process): six at 512^3, three at 256^3.
cd "$(dirname "$0")"
[ -x ./streams ] || hipcc --offload-arch=gfx950 -O3 streams.hip -o streams
for pad in 0 0 3 37 517 2049; do ./streams 512 $pad 0; done
for pad in 0 5 131; do ./streams 256 $pad 12288; done
