cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_unet_hip_parity.py -q -x 2>&1 | tail -1
timeout -k 10 300 python tools/bench_unet.py --steps 10 --warmup 2 2>/dev/null | cut -c1-160
