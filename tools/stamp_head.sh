#!/bin/bash
# record HEAD beside the built libraries (see bench.py git_head): run before a gpurun call whose outputs are kept
cd "$(dirname "$0")/.." && h=$(git rev-parse --short=12 HEAD) && if git diff --quiet HEAD -- staticfusion_amd include bench.py; then echo $h; else echo $h+dirty; fi > staticfusion_amd/csrc/BUILD_HEAD && cat staticfusion_amd/csrc/BUILD_HEAD
