f() { python bench.py "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms  kernel %.4f  windows %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], [round(x,2) for x in d['windows_ms']['all']]))"; }
echo "A steps30 nogather nocpu"; f --steps 30 --warmup 5 --cpu-steps 0 --no-gather
echo "B steps50 nogather nocpu"; f --steps 50 --warmup 5 --cpu-steps 0 --no-gather
echo "C steps50 gather nocpu";   f --steps 50 --warmup 5 --cpu-steps 0
echo "D steps50 warm50";         f --steps 50 --warmup 50 --cpu-steps 0 --no-gather
echo "E steps100";               f --steps 100 --warmup 5 --cpu-steps 0 --no-gather
echo "A again";                  f --steps 30 --warmup 5 --cpu-steps 0 --no-gather
