"""Study (not a test): cost of the one-shot `.Call` payload at BASELINE config 3 -- what the R host pays per
euclidean_embedding() call besides the relaxation itself (session creation, verification of the inputs, upload,
encode, download).  TOPOLOW_DENSE_UPLOAD=1 forces round 1's path (upload of the dense 800 + 400 MB)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import parity_problems as pp
from tests.conftest import layout_call_args
from topolow_amd import _native

call, _ = pp.cfg3_generator(10000)
rows = []
for rep in range(4):
    t0 = time.perf_counter()
    r = _native.optimize_layout_exact_arrays(*layout_call_args(call), seed=1 + rep)
    wall = time.perf_counter() - t0
    i = r.info
    rows.append(dict(wall=wall, total=i["total_seconds"], setup=i["setup_seconds"], device=i["device_seconds"],
                     iterations_run=i["iterations_run"], final_mae=r.final_mae))
    print(f"call {rep}: wall {wall:.3f} s, library total {i['total_seconds']:.3f} s = setup {i['setup_seconds']:.3f} + "
          f"relaxation {i['device_seconds']:.3f} ({i['iterations_run']} iterations) + download/teardown "
          f"{i['total_seconds'] - i['setup_seconds'] - i['device_seconds']:.3f}; total - device = "
          f"{i['total_seconds'] - i['device_seconds']:.3f} s", flush=True)
json.dump(dict(dense_upload=os.environ.get("TOPOLOW_DENSE_UPLOAD", "0"), calls=rows), open(sys.argv[1], "w"))
