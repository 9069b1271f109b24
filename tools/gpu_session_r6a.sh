mkdir -p gpurun_out/r6a
python bench.py > gpurun_out/r6a/bench_N1.json 2> gpurun_out/r6a/bench_N1.err; echo "rc default $?"
python bench.py --total-perms 1073741824 --steps 3 --warmup 1 --no-secondary > gpurun_out/r6a/bench_strong_2p30_N1.json 2> gpurun_out/r6a/bench_strong.err; echo "rc strong $?"
python bench.py --gpus 2 --single-device --dist-backend gloo --total-perms 4194304 --steps 2 --warmup 1 --cpu-sample 4096 > gpurun_out/r6a/bench_strong_rehearsal_2ranks.json 2> gpurun_out/r6a/bench_strong2.err; echo "rc strong2 $?"
python bench.py --gpus 8 --single-device --dist-backend gloo --perms-per-gpu 1048576 --steps 2 --warmup 1 --cpu-sample 4096 > gpurun_out/r6a/bench_rehearsal_8ranks.json 2> gpurun_out/r6a/bench_r8.err; echo "rc r8 $?"
timeout 1200 python -m pytest tests/test_bench_contract.py tests/test_gpu_e_multigpu.py -m gpu -x -q > gpurun_out/r6a/pytest_bench.txt 2>&1; echo "rc pytest $?"
tail -5 gpurun_out/r6a/pytest_bench.txt
