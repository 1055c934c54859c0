mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/t22_all.log 2>&1; echo "all gpu tests rc=$?"; tail -n 4 gpurun_out/r03/t22_all.log
FYPRT_SEQ_FIRST=6000 FYPRT_SEQ_LAST=6120 timeout -k 10 300 python -m pytest tests/test_gpu_api_sequences.py -m gpu -x -q > gpurun_out/r03/soak_api2.log 2>&1; echo "soak api rc=$?"; tail -n 2 gpurun_out/r03/soak_api2.log
FYPRT_SEQ_FIRST=6000 FYPRT_SEQ_LAST=6080 timeout -k 10 300 python -m pytest tests/test_gpu_group_sequences.py -m gpu -x -q > gpurun_out/r03/soak_group2.log 2>&1; echo "soak group rc=$?"; tail -n 2 gpurun_out/r03/soak_group2.log
