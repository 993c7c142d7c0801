"""The two roles of the multi-GPU job alone on one GPU (bench.py:role_timings), with the host-side figures (SRLX_ROLE_PROBE)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SRLX_ROLE_PROBE"] = "1"
import bench
sys.argv = [sys.argv[0]] + sys.argv[1:]
args = bench.parse_args()
print(json.dumps(bench.role_timings(args, 0), indent=1))  # (no RCCL in this process; `python bench.py --roles-only` initialises a one-rank group first)
