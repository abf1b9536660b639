#!/bin/bash
# block -> CU placement probe (run on the MI355X box): compiles tools/micro/cu_map.hip and runs it
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o /tmp/cu_map cu_map.hip && /tmp/cu_map
