for cfg in c5 c4; do for P in 1 2 3 1 2 3; do
ms=$(DTSIM_RENDER_PARTS=$P python bench.py --config $cfg --steps 20 --warmup 3 --windows 3 --cpu-steps 0 --no-gather 2>/dev/null | tail -1 | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
echo "$cfg parts=$P $ms"; done; done
