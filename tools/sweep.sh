run() { python bench.py --steps $1 --warmup 20 --no-cpu-baseline --no-served-legs --no-flip-rate --no-roofline "${@:2}" 2>&1 | grep "^{" | grep -o "\"value\": [0-9.]*\|\"ms_per_step\": [0-9.]*" | tr "\n" " "; echo " :: ${@:2}"; }
run 200 --batch 2 --streams 4
run 200 --batch 2 --streams 2
run 100 --batch 4 --streams 2
run 100 --batch 4 --streams 3
run 60 --batch 8 --streams 2
run 30 --batch 28 --streams 1
run 30 --batch 28 --streams 2
