cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/scripts_e2e.sh r04q_e2e > gpurun_out/r04q_e2e.txt 2>&1
export OSVOS_SAVE_ROOT=/tmp/pp OSVOS_MODELS_DIR=/tmp/pp; mkdir -p /tmp/pp
echo "== train_parent.py --synthetic 512 --epochs 3 (fp32x3; frames as host tensors, no device input pipeline)" >> gpurun_out/r04q_e2e.txt
timeout 600 python train_parent.py --synthetic 512 --epochs 3 2>&1 | grep "Execution" | tr '\n' ' ' >> gpurun_out/r04q_e2e.txt
timeout 900 python -m pytest tests/test_gpu_scripts.py tests/test_gpu_augment.py -x -q 2>&1 | tail -3 >> gpurun_out/r04q_e2e.txt
