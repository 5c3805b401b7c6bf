# on-box: is the training step launch-bound?  wall per step under different queue / kernarg settings
for q in 4 12; do for k in 0 1; do
echo "GPU_MAX_HW_QUEUES=$q HIP_FORCE_DEV_KERNARG=$k"
GPU_MAX_HW_QUEUES=$q HIP_FORCE_DEV_KERNARG=$k python tools/bench_train.py --dtype bf16 --no-cpu-baseline 2>&1 | tail -1 | cut -c80-170
done; done
