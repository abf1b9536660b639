run() { python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-roofline "$@" 2>&1 | tail -1 | grep -o "\"value\": [0-9.]*\|\"ms_per_step\": [0-9.]*\|host_post_ms_per_frame\": [0-9.]*" | tr "\n" " "; echo " :: $@"; }
run --streams 4
run --streams 6
run --streams 8
run --streams 6 --sk-target 256
run --streams 6 --sk-target 256 --sk-min 8
run --streams 6 --sk-target 128 --sk-min 8 --sk-max 4
run --streams 6 --sk-target 1 
run --streams 6 --tile 1
run --streams 6 --tile 1 --sk-target 256 --sk-min 8
run --streams 8 --sk-target 256 --sk-min 8
