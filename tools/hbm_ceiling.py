#!/usr/bin/env python3
"""Measured HBM ceilings of the box next to the vendor figure the roofline uses (8 TB/s):
device-to-device copy (read + write) and a stream triad a = b + s*c (two reads + one write), on buffers far larger
than the 256 MB memory-side cache.  torch supplies the buffers and the two elementwise launches; nothing of the
echo chain runs here.  Prints one JSON line."""
import json
import sys

import torch


def timed(fn, iters):
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / iters / 1e3


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    n = int(gib * (1 << 30)) // 4
    dev = torch.device("cuda:0")
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.ones(n, dtype=torch.float32, device=dev)
    c = torch.ones(n, dtype=torch.float32, device=dev)
    t_copy = timed(lambda: a.copy_(b), 20)
    t_triad = timed(lambda: torch.add(b, c, alpha=0.5, out=a), 20)
    nbytes = n * 4
    print(json.dumps({
        "device": torch.cuda.get_device_name(0), "buffer_GiB": gib,
        "copy_GBps": 2 * nbytes / t_copy / 1e9, "triad_GBps": 3 * nbytes / t_triad / 1e9,
        "vendor_peak_GBps": 8000.0,
        "note": "bytes moved = reads + writes; buffers exceed the memory-side cache"}))


if __name__ == "__main__":
    main()
