mkdir -p gpurun_out/r6e
timeout 900 python -m pytest tests/test_gpu_f4_witness.py tests/test_gpu_a13_fr.py tests/test_gpu_f1_sponge.py -m gpu -x -q > gpurun_out/r6e/pytest.txt 2>&1; echo "rc $?"; tail -5 gpurun_out/r6e/pytest.txt
