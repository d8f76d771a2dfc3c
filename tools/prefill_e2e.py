"""End-to-end prefill (time to first token) of Llama-3-8B through this package's model -- bench.py's prefill_e2e section on its
own (for rocprofv3 --kernel-trace --stats):  python tools/prefill_e2e.py [T] [int4|fp8|both] [--library]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
which = sys.argv[2] if len(sys.argv) > 2 else "both"
which = ("int4", "fp8") if which == "both" else (which, )
print(json.dumps(bench.prefill_e2e_section(T, library="--library" in sys.argv, which=which)))
