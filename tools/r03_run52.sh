mkdir -p gpurun_out/r03
FYPRT_SEQ_FIRST=9000 FYPRT_SEQ_LAST=9220 timeout -k 10 520 python -m pytest tests/test_gpu_api_sequences.py -m gpu -x -q > gpurun_out/r03/soak_api5.log 2>&1; echo "soak api rc=$?"; tail -n 2 gpurun_out/r03/soak_api5.log
FYPRT_SEQ_FIRST=9000 FYPRT_SEQ_LAST=9250 timeout -k 10 400 python -m pytest tests/test_gpu_group_sequences.py -m gpu -x -q > gpurun_out/r03/soak_group5.log 2>&1; echo "soak group rc=$?"; tail -n 2 gpurun_out/r03/soak_group5.log
