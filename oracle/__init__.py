"""TEST INFRASTRUCTURE: CPU (fp32 PyTorch) restatement of the reference hot path. See unet_ref.py."""
