#!/usr/bin/env python3
"""What can this box's host link do?  Pinned-memory DMA through the C ABI (gc_host_alloc + gc_dev_upload /
gc_dev_download on two contexts), each direction alone and both at once.  Reference for bench_host_api.py."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.bench_host_api import link_probe

if __name__ == "__main__":
    print(json.dumps(link_probe(int(sys.argv[1]) if len(sys.argv) > 1 else 256 << 20)))
