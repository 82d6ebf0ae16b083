"""write the native driver's program file of one of bench_stream's programs:  write_program.py NAME PATH [WINDOW]
(then: tools/stream_driver PATH — e.g. under rocprofv3)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_stream as bs  # noqa: E402

name, path = sys.argv[1], sys.argv[2]
window = int(sys.argv[3]) if len(sys.argv) > 3 else (2 if name.startswith("big") else 64)
steps, prim = bs.PROGRAMS[name]()
bs.write_program(path, bytes(range(32)), bs.stream_rnd(name, len(prim)), prim, steps, window)
