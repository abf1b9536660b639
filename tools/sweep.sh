run() { python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-roofline "$@" 2>&1 | grep "^{" | grep -o "\"value\": [0-9.]*\|\"ms_per_step\": [0-9.]*" | tr "\n" " "; echo " :: $QQ $@"; }
QQ=""; run --streams 4
QQ=""; run --streams 4 --sk-target 256
QQ=""; run --streams 4 --sk-target 384 --sk-min 6
QQ=""; run --streams 4 --sk-target 768
QQ=""; run --streams 4 --sk-max 4
QQ=""; run --streams 4 --sk-max 16
export GPU_MAX_HW_QUEUES=8; QQ="HWQ8"
run --streams 4
run --streams 6
run --streams 8
export GPU_MAX_HW_QUEUES=16; QQ="HWQ16"
run --streams 8
run --streams 12
