cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04t; mkdir -p $O
( export OSVOS_SAVE_ROOT=/tmp/pp OSVOS_MODELS_DIR=/tmp/pp; mkdir -p /tmp/pp
  for q in 4 8; do for p in fp32x3 bf16; do echo "== GPU_MAX_HW_QUEUES=$q train_parent.py --synthetic 512 --device-augment --epochs 3 --precision $p"; GPU_MAX_HW_QUEUES=$q timeout 600 python train_parent.py --synthetic 512 --device-augment --epochs 3 --precision $p 2>&1 | grep "Execution" | tr '\n' ' '; echo; done; done
  for q in 4 8; do echo "== GPU_MAX_HW_QUEUES=$q OSVOS_DP_FORCE=1 (RCCL communicator live)"; MASTER_ADDR=127.0.0.1 MASTER_PORT=29813 OSVOS_DP_FORCE=1 GPU_MAX_HW_QUEUES=$q timeout 600 python train_parent.py --synthetic 512 --device-augment --epochs 3 2>&1 | grep "Execution" | tr '\n' ' '; echo; done ) > $O/queues.txt 2>&1
timeout 900 python -m pytest tests/test_augment.py tests/test_gpu_scripts.py -x -q 2>&1 | tail -3 > $O/pytest.txt
bash tools/scripts_e2e.sh r04t_e2e > $O/scripts_e2e.txt 2>&1
bash tools/gpu_pmc_step.sh r04t_pmc1 "configs[1] 854x480 b1 online f32x3" > $O/pmc1.txt 2>&1
bash tools/gpu_pmc_step.sh r04t_pmc2 "configs[2] 854x480 b12 parent bf16" --mode parent --precision bf16 --batch 12 > $O/pmc2.txt 2>&1
