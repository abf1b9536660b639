#!/bin/bash
# LDS bandwidth micro-benchmark (run on the MI355X box): compiles tools/micro/lds_bw.hip and prints B/clk/CU
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_bw lds_bw.hip && /tmp/lds_bw
