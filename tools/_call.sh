mkdir -p gpurun_out/c2
timeout -k 10 600 python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_sharded_api.py -x -q > gpurun_out/c2/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/c2/tests.log
timeout -k 10 400 python tools/pcie_probe.py > gpurun_out/c2/pcie_probe.txt 2>&1; cat gpurun_out/c2/pcie_probe.txt | grep -v amdgpu.ids
nproc
