#!/bin/bash
# block -> XCD mapping probe (run on the MI355X box): compiles tools/micro/xcc_map.hip and runs it
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o /tmp/xcc_map xcc_map.hip && /tmp/xcc_map
