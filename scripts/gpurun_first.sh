cd $GRAFT_REPO_ROOT
python -c "import torch; print(torch.cuda.is_available(), torch.cuda.get_device_name(0))"
timeout 600 python -m pytest tests/test_scan_gpu.py -x -q -m gpu 2>&1 | tail -30
