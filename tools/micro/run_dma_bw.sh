#!/bin/bash
# LDS-DMA throughput micro-benchmark (run on the MI355X box): compiles tools/micro/dma_bw.hip and prints B/clk/CU
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/dma_bw dma_bw.hip && /tmp/dma_bw
