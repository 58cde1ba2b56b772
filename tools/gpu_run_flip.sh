cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/h2
python tools/net_flip_probe.py > gpurun_out/h2/net_flips.txt 2>&1; grep -v Constructing gpurun_out/h2/net_flips.txt | grep -v Initializing | tail -20
python tools/net_flip_probe.py 480 854 > gpurun_out/h2/net_flips_480.txt 2>&1; grep -v Constructing gpurun_out/h2/net_flips_480.txt | grep -v Initializing | tail -16
