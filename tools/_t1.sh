mkdir -p gpurun_out/t2
python -m pytest tests/test_backward_gpu.py tests/test_abi.py -q -x -m gpu > gpurun_out/t2/pytest.txt 2>&1
python -m pytest tests/test_model_gpu.py -q -x -k "train or grad" >> gpurun_out/t2/pytest.txt 2>&1
python bench.py --mode train --steps 5 --warmup 2 > gpurun_out/t2/train.json 2> gpurun_out/t2/train.err
