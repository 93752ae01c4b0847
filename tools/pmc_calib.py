#!/usr/bin/env python3
"""Known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this box (MI355X_MICROARCH.md §HBM:
FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950; other widths uncalibrated): a 4-B-per-lane stream
(cmtts transpose_kernel: reads N floats, writes N floats) and a 16-B-per-lane stream (torch's vectorised elementwise
copy-scale), 256 MiB each way, well past the 256 MiB Infinity Cache in total."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cmtts_amd
from cmtts_amd import host

x = torch.randn(64, 1024, 1024, device="cuda")          # 256 MiB
for _ in range(3):
    y = host.transpose_last2(x)                         # transpose_kernel: 4 B per lane both ways
    z = x * 1.0001                                      # vectorized_elementwise_kernel: 16 B per lane both ways
torch.cuda.synchronize()
print("bytes each way per launch:", x.numel() * 4)
