set -o pipefail
bash scripts/profile.sh r01h_1080p8 > gpurun_out/prof_r01h_1080p8.log 2>&1
BENCH_FLAGS="--width 3840 --rows-per-gpu 270 --spp 32" bash scripts/profile.sh r01h_4k32slab > gpurun_out/prof_r01h_4k32slab.log 2>&1
python bench.py > gpurun_out/bench_r01h.json 2>gpurun_out/bench_r01h.err
python bench.py --width 3840 --rows-per-gpu 270 --spp 32 --steps 3 --warmup 1 > gpurun_out/bench_r01h_4k32slab.json 2>gpurun_out/bench_r01h_4k32slab.err
bash scripts/ablate.sh > gpurun_out/ablate_r01h.txt 2>&1
bash scripts/sweep.sh > gpurun_out/sweep_r01h.txt 2>&1
python scripts/host_path.py 2>&1 | grep -v amdgpu > gpurun_out/host_path_r01h.txt
(python scripts/smalln.py; SPP=16 python scripts/smalln.py; SPP=32 ROWS=540 python scripts/smalln.py; SPP=64 ROWS=270 python scripts/smalln.py) 2>&1 | grep -v amdgpu > gpurun_out/smalln_r01h.txt
(python scripts/multipass.py 8; python scripts/multipass.py 16; python scripts/multipass.py 32) 2>&1 | grep -v amdgpu > gpurun_out/multipass_r01h.txt
tail -2 gpurun_out/multipass_r01h.txt
