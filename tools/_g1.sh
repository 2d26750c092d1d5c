cd /root/repo
python tools/ab.py --reps 3 "TN_SK_RB=16" "TN_SK_RB=8" 
TN_SK_RB=8 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "softmax_train" 2>&1 | tail -2
