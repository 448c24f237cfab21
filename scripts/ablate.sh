# stage-time ablation of the fused kernel (option stage_mask is a profiling knob; outputs are wrong when set)
# chain = t(1)-t(0); bins = t(3)-t(1); MI = t(7)-t(3); weights = t(-1)-t(7); t(0) = gather + launch
for m in -1 0 1 3 7; do python bench.py --option stage_mask=$m --steps 2 --warmup 1 --no-cpu-baseline $ABLATE_FLAGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mask', $m, 'kernel_ms %.1f'%d['roofline']['kernel_ms'], 'ms_per_step %.1f'%d['ms_per_step'])"; done
