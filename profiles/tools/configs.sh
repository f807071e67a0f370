set -u
run() { python bench.py "$@" --no-cpu-baseline --instrumented-steps 0 --no-exclusive 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1:], d['value'], d['ms_per_step'], d.get('value_resident'), d['config']['sharding'])" "$@"; }
run --cells 50000 --genes 20000 --density 0.05
run --algorithm louvain --scaling
run --algorithm leiden
run --cells 500000 --genes 33000 --density 0.02 --steps 4 --warmup 2
rocm-smi --showmeminfo vram | grep Used || true
