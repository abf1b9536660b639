#!/bin/bash
# first-touch fetch rates per CU by where the data is (run on the MI355X box): compiles tools/micro/cold_fetch.hip and runs it
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/cold_fetch cold_fetch.hip && /tmp/cold_fetch
